import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "nmf.jl_amd"))
import nmfx
os.environ["NMFX_DEV"] = "1"
T = np.float32; p = n = 16384; k = 256
rng = np.random.default_rng(5)
X = np.asfortranarray((rng.random((p, k), dtype=T) @ rng.random((k, n), dtype=T)).astype(T))
W0 = np.asfortranarray(rng.random((p, k), dtype=T)); H0 = np.asfortranarray(rng.random((k, n), dtype=T))
with nmfx.Context(T, p, n, k) as ctx:
    ctx.set_X(X)
    W, H = W0.copy(order="F"), H0.copy(order="F")
    ctx.solve(0, nmfx.make_opts(T, maxiter=30, tol=1e-30, exact_stop=True, update_H=False), W, H)
