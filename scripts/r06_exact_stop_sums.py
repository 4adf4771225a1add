#!/usr/bin/env python3
"""Round 6: cost of nmfx_opts.stop_sums = 1 (the reference's sequential T-precision stop sums) at the headline shape, second form of the
kernel against the first (NMFX_STOP_SUMS_V1=1) and against the default tree sums; identical relchange columns between the two forms."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "nmf.jl_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import nmfx  # noqa: E402

os.environ["NMFX_DEV"] = "1"
T = np.float32
p = n = 16384
k = 256
rng = np.random.default_rng(5)
X = np.asfortranarray((rng.random((p, k), dtype=T) @ rng.random((k, n), dtype=T)).astype(T))
W0 = np.asfortranarray(rng.random((p, k), dtype=T))
H0 = np.asfortranarray(rng.random((k, n), dtype=T))
rc = {}
for name, env, exact in (("tree sums (default)", "0", False), ("exact, second form", "0", True), ("exact, first form", "1", True)):
    os.environ["NMFX_STOP_SUMS_V1"] = env
    with nmfx.Context(T, p, n, k) as ctx:
        ctx.set_X(X)
        best = 1e9
        for rep in range(3):
            W, H = W0.copy(order="F"), H0.copy(order="F")
            t0 = time.perf_counter()
            res, _ = ctx.solve(0, nmfx.make_opts(T, maxiter=200, tol=1e-30, exact_stop=exact), W, H)
            best = min(best, (time.perf_counter() - t0) / res.niters)
        W, H = W0.copy(order="F"), H0.copy(order="F")
        ctx.solve(0, nmfx.make_opts(T, maxiter=6, tol=1e-30, exact_stop=exact, track_objective=True), W, H)
        _, r = ctx.iter_trace(7)
        rc[name] = np.array(r[1:7])
    print(f"{name:22s} {best * 1e3:.4f} ms per iteration (wall clock of a 200-iteration solve incl. up/download, best of 3)", flush=True)
a, b = rc["exact, second form"], rc["exact, first form"]
print("relchange columns of the two exact forms identical:", bool(np.array_equal(a, b)), a[:3], b[:3], rc["tree sums (default)"][:3])
