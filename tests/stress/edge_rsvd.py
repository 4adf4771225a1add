"""edge cases of the device rsvd / nndsvd (run by hand on the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for sub in ("nmf.jl_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import numpy as np, nmfx
rng = np.random.default_rng(0)
for T in (np.float32, np.float64):
    for (p, n, k) in [(3, 5, 3), (5, 3, 1), (1, 1, 1), (40, 2000, 2), (2000, 40, 40), (300, 300, 1)]:
        X = np.asfortranarray(rng.random((p, n)).astype(T))
        U, s, V = nmfx.rsvd(X, k, seed=1, power_iters=2)
        err = np.linalg.norm(X - (U * s) @ V.T)
        sv = np.linalg.svd(X.astype(np.float64), compute_uv=False)
        best = np.sqrt((sv[k:] ** 2).sum())
        W, H = nmfx.nndsvd(X, k, variant="ar", seed=2, power_iters=1)
        r = nmfx.nnmf(X, k, init="nndsvda", alg="greedycd", maxiter=5)
        print(np.dtype(T).name, (p, n, k), "rsvd err %.3e optimal %.3e" % (err, best), "orth %.1e" % np.abs(U.T @ U - np.eye(k)).max(),
              "nndsvd nonneg", bool((W >= 0).all() and (H >= 0).all()), "nnmf objv %.4e" % r.objvalue, flush=True)
    Z = np.zeros((20, 30), dtype=T, order="F")
    U, s, V = nmfx.rsvd(Z, 2, seed=1)
    print(np.dtype(T).name, "zero X: s =", s, "finite U,V:", bool(np.isfinite(U).all() and np.isfinite(V).all()))
