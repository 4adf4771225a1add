"""Development aid (GPU box): wall time and device-loop time of nmfx_iterate for 1 ... 100 iterations of the headline workload, and the
first call after 2 s of idle -- the per-call overhead and the clock ramp behind bench.py's `--prewarm-ms` and the deferred final
objective (profiles/r04_iterate_call_overhead.log)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nmf.jl_amd"))
import numpy as np, torch, nmfx, bench
p = n = 16384; k = 256
dev = torch.device("cuda:0")
Xt, W0, H0 = bench.synth(p, n, k, 0, n, torch.float32, dev)
with nmfx.Context(np.float32, p, n, k) as ctx:
    ctx.set_X_device(Xt.data_ptr(), p); ctx.set_factors(W0, H0)
    def it(m):
        o = nmfx.make_opts(np.float32, maxiter=m, tol=1e-38, check_every=1000)
        torch.cuda.synchronize(); t0 = time.perf_counter(); r, _ = ctx.iterate(0, o); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3, r.seconds_loop * 1e3
    it(5)
    for m in (1, 2, 5, 10, 20, 50, 100, 20, 20, 5, 1):
        w, l = it(m)
        print(f"steps {m:4d}: wall {w:8.3f} ms ({w/m:.3f}/step)  device loop {l:8.3f} ms ({l/m:.3f}/step)  wall - loop {w-l:.3f}", flush=True)
    time.sleep(2.0)
    for m in (20, 20):
        w, l = it(m); print(f"after 2 s idle: steps {m}: wall {w:.3f} ({w/m:.3f}/step) loop {l:.3f} ({l/m:.3f})")
