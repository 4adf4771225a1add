"""extra shape sweep (cd / greedycd dispatch paths, odd k)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for sub in ("nmf.jl_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import numpy as np
import nmfx, nmf_oracle as orc, c_oracle as co
from problems import uniform, rel_trace_err
SHAPES = [(700, 600, 64), (700, 600, 65), (600, 700, 128), (650, 700, 129), (900, 800, 300), (1200, 1100, 513), (2000, 70, 1), (70, 2000, 2), (515, 513, 33)]
bad = 0
for (p, n, k) in SHAPES:
    for T in (np.float32, np.float64):
        for alg in ("cd", "greedycd", "multmse", "projals"):
            if alg == "greedycd" and k > 130: continue
            X, W0, H0 = uniform(p, n, k, T, seed=p + n + k)
            if alg == "projals": W0 = np.asfortranarray(np.random.default_rng(1).random((p, k)).astype(T))
            iters = 3
            lam = 0.5 if alg == "projals" else 0.0
            inst = {"cd": lambda: nmfx.CoordinateDescent(T, maxiter=iters, tol=1e-30, alpha=1e-3, l1ratio=0.5),
                    "greedycd": lambda: nmfx.GreedyCD(T, maxiter=iters, tol=1e-30),
                    "multmse": lambda: nmfx.MultUpdate(T, maxiter=iters, tol=1e-30),
                    "projals": lambda: nmfx.ProjectedALS(T, maxiter=iters, tol=1e-30, lambda_w=lam, lambda_h=lam)}[alg]()
            W, H = W0.copy(order="F"), H0.copy(order="F")
            try:
                r = nmfx.solve(inst, X, W, H, track_objective=True)
            except Exception as e:
                print(f"{p}x{n} k={k} {np.dtype(T).name} {alg}: EXC {type(e).__name__}: {e}"); bad += 1; continue
            kw = dict(l1_w=inst.l1_w, l2_w=inst.l2_w, l1_h=inst.l1_h, l2_h=inst.l2_h) if alg == "cd" else dict(lambda_w=lam, lambda_h=lam)
            ro = co.solve(alg, X, W0.copy(order="F"), H0.copy(order="F"), orc.Opts(maxiter=iters, tol=1e-30, track_objective=True, **kw))
            err = rel_trace_err(r.trace, ro.trace)
            tol = {np.float32: 2e-1 if alg == "greedycd" else 2e-3, np.float64: 1e-7}[T]
            ok = err < tol and np.all(W >= 0) and np.all(H >= 0)
            bad += (not ok)
            print(f"{p}x{n} k={k} {np.dtype(T).name} {alg}: err {err:.2e} {'ok' if ok else 'FAIL'}", flush=True)
print("FAILURES:", bad)
