"""Shape sweep on the GPU box: every algorithm, both dtypes, awkward sizes, compared with the NumPy oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for sub in ("nmf.jl_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import numpy as np
import nmfx, nmf_oracle as orc, c_oracle as co
from problems import uniform, rel_trace_err

SHAPES = [(1, 1, 1), (2, 3, 1), (7, 5, 5), (33, 1000, 3), (1000, 33, 17), (129, 257, 128), (129, 257, 100), (256, 300, 256), (640, 300, 129), (300, 20000, 20),
          (20000, 300, 20), (520, 530, 500), (1100, 1200, 1000)]
bad = 0
for (p, n, k) in SHAPES:
    for T in (np.float32, np.float64):
        for alg in ("multmse", "multdiv", "projals", "alspgrad", "cd", "greedycd"):
            if alg in ("alspgrad", "greedycd") and k > 200:
                continue
            X, W0, H0 = uniform(p, n, k, T, seed=p + n + k)
            iters = 3 if alg == "alspgrad" else 5
            lam = 0.5 if alg == "projals" else 1e-4
            if alg in ("multmse", "multdiv"):
                inst = nmfx.MultUpdate(T, obj=alg[4:], maxiter=iters, tol=1e-30, lambda_w=lam, lambda_h=lam)
            elif alg == "projals":
                inst = nmfx.ProjectedALS(T, maxiter=iters, tol=1e-30, lambda_w=lam, lambda_h=lam)
            elif alg == "cd":
                inst = nmfx.CoordinateDescent(T, maxiter=iters, tol=1e-30)
            elif alg == "greedycd":
                inst = nmfx.GreedyCD(T, maxiter=iters, tol=1e-30, lambda_w=lam, lambda_h=lam)
            else:
                inst = nmfx.ALSPGrad(T, maxiter=iters, tol=1e-30, maxsubiter=15)
            W, H = W0.copy(order="F"), H0.copy(order="F")
            t0 = time.time()
            try:
                r = nmfx.solve(inst, X, W, H, track_objective=True)
            except Exception as e:
                print(f"{p}x{n} k={k} {np.dtype(T).name} {alg}: EXC {type(e).__name__}: {e}")
                if not (alg == "projals" and "too large" in str(e)):
                    bad += 1
                continue
            tg = time.time() - t0
            Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
            o = orc.Opts(maxiter=iters, tol=1e-30, lambda_w=lam, lambda_h=lam, track_objective=True, maxsubiter=15)
            ro = (co if alg in ("cd", "greedycd") else orc).solve(alg, X, Wc, Hc, o)
            if len(r.trace) != len(ro.trace):
                print(f"{p}x{n} k={k} {np.dtype(T).name} {alg}: niters {r.niters}/{ro.niters} converged {r.converged}/{ro.converged} trace {r.trace} vs {ro.trace} FAIL", flush=True)
                bad += 1
                continue
            err = rel_trace_err(r.trace, ro.trace)
            loose = alg in ("projals", "alspgrad", "cd", "greedycd")
            tol = {np.float32: (2e-1 if alg == "greedycd" else 5e-4) if loose else 2e-5, np.float64: 1e-7 if loose else 1e-10}[T]
            ok = err < tol and r.niters == ro.niters and np.all(W >= 0) and np.all(H >= 0)
            if not ok:
                bad += 1
            print(f"{p}x{n} k={k} {np.dtype(T).name} {alg}: niters {r.niters}/{ro.niters} objective rel err {err:.2e} {'ok' if ok else 'FAIL'} ({tg:.2f}s)", flush=True)
print("FAILURES:", bad)
