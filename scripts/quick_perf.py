"""Quick per-kernel timing on the GPU box (development aid, not the bench contract)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nmf.jl_amd"))
import numpy as np
import nmfx

def run(p, n, k, T, alg_name, iters=10, maxsub=200, comm=False):
    rng = np.random.default_rng(0)
    t0 = time.time()
    Wg = rng.random((p, k), dtype=np.float32); Hg = rng.random((k, n), dtype=np.float32)
    X = np.asfortranarray(rng.random((p, n), dtype=np.float32).astype(T)) if alg_name == "projals" else np.asfortranarray((Wg @ Hg).astype(T))
    W0 = rng.random((p, k)).astype(T)
    if alg_name != 'projals': W0 /= W0.sum(0, keepdims=True)
    W0 = np.asfortranarray(W0)
    H0 = np.asfortranarray(rng.random((k, n)).astype(T))
    print(f"gen {time.time()-t0:.1f}s", flush=True)
    algs = {"multmse": 0, "multdiv": 1, "projals": 2, "alspgrad": 3, "cd": 4, "greedycd": 5}
    with nmfx.Context(T, p, n, k) as ctx:
        t0 = time.time(); ctx.set_X(X); print(f"upload X {time.time()-t0:.2f}s", flush=True)
        if comm:
            ctx.comm_init(nmfx.api.comm_unique_id(), 0, 1)   # 1-rank RCCL communicator: exercises the sharded code path
        ctx.set_factors(W0, H0)
        lam = {"multdiv": 3.5e-4, "projals": 0.5}.get(alg_name, 0.0)
        o = nmfx.make_opts(T, maxiter=3, tol=1e-30, check_every=1000, lambda_w=lam, lambda_h=lam, maxsubiter=maxsub)
        ctx.iterate(algs[alg_name], o)  # warmup
        o.maxiter = iters
        t0 = time.time(); res, _ = ctx.iterate(algs[alg_name], o); wall = time.time() - t0
        fl = {"multmse": 4.0*p*n*k + 4.0*k*k*(p+n), "multdiv": 8.0*p*n*k,
              "projals": 4.0*p*n*k + 2.0*k*k*(p+n) + 2.0*k*k*n + 2.0*p*k*k + float(k)**3,
              "alspgrad": 4.0*p*n*k + 2.0*k*k*(p+n), "cd": 4.0*p*n*k + 4.0*k*k*(p+n), "greedycd": 4.0*p*n*k + 4.0*k*k*(p+n)}[alg_name]
        print(f"   inner={res.inner_iters} backtracks={res.backtracks}")
        print(f"{alg_name} {p}x{n} k={k} {np.dtype(T).name}: {res.seconds_loop/iters*1e3:.3f} ms/iter (wall {wall/iters*1e3:.3f}) "
              f"-> {fl*iters/res.seconds_loop/1e12:.1f} TFLOP/s alg; objv {res.objvalue:.6e}", flush=True)
        ctx.profile_enable(True)
        ctx.iterate(algs[alg_name], o)
        for s in ctx.profile_get():
            ms = s["ms_total"] / max(1, s["launches"])
            tf = s["flops"] / max(1, s["launches"]) / (ms * 1e-3) / 1e12 if s["flops"] else 0
            gb = s["bytes"] / max(1, s["launches"]) / (ms * 1e-3) / 1e9 if s["bytes"] else 0
            print(f"   {s['name']:<22s} n={s['launches']:<4d} avg {ms*1e3:9.1f} us  {tf:7.1f} TF/s {gb:8.0f} GB/s")

if __name__ == "__main__":
    which = sys.argv[1:] or ["c2", "c3"]
    if "c2" in which: run(4096, 4096, 64, np.float32, "multmse", 20)
    if "c3" in which: run(16384, 16384, 256, np.float32, "multmse", 10)
    if "c3div" in which: run(16384, 16384, 256, np.float32, "multdiv", 5)
    if "c3f64" in which: run(8192, 8192, 256, np.float64, "multmse", 5)
    if "c4shard" in which: run(16384, 16384, 256, np.float32, "projals", 5)
    if "c5shard" in which: run(8192, 4096, 512, np.float64, "alspgrad", 2, maxsub=10)
    if "cd" in which: run(16384, 16384, 256, np.float32, "cd", 10)
    if "gcd" in which: run(16384, 16384, 256, np.float32, "greedycd", 5)
    if "cd2" in which: run(4096, 4096, 64, np.float32, "cd", 20); run(4096, 4096, 64, np.float32, "greedycd", 10)
    if "c5full" in which: run(32768, 4096, 512, np.float64, "alspgrad", 2, maxsub=10)
    if "shards" in which:      # per-rank shapes of the C3 problem at 2/4/8 GPUs (local compute + 1-rank all-reduce)
        for nl in (8192, 4096, 2048): run(16384, nl, 256, np.float32, "multmse", 30, comm=True)
    if "alsf32" in which: run(4096, 4096, 64, np.float32, "alspgrad", 2, maxsub=20)
