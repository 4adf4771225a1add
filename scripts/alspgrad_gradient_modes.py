#!/usr/bin/env python
"""ALSPGrad: the exact gradient (G = Gram*Z - B by a full product every inner iteration, src/alspgrad.jl:124-127, 280-283;
nmfx_opts.pg_refresh = 1) against the running form (G += Gram*D of the accepted step, full product every n-th iteration).

Parity: objective trajectory and the inner-iteration / back-tracking counters against the CPU oracle on seeded problems, both element
types.  Time: one shape per element type on the device (outer iterations of the full solver).  Prints one JSON object per line.
usage (GPU box): python scripts/alspgrad_gradient_modes.py > gpurun_out/alspgrad_gradient_modes.jsonl"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("nmf.jl_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import torch  # noqa: E402,F401
import nmfx  # noqa: E402
import nmf_oracle as orc  # noqa: E402
from problems import planted, rel_trace_err  # noqa: E402

MODES = [("exact", "exact"), ("refresh16", 16), ("refresh64", 64)]


def parity(T, shape, seed, iters=8):
    p, n, k = shape
    X, W0, H0 = planted(p, n, k, T, seed=seed)
    Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
    ro = orc.solve("alspgrad", X, Wc, Hc, orc.Opts(maxiter=iters, tol=1e-30, track_objective=True))
    row = {"what": "parity", "dtype": np.dtype(T).name, "shape": list(shape), "seed": seed, "outer_iterations": iters,
           "oracle_inner": ro.counters["inner"], "oracle_backtracks": ro.counters["backtracks"]}
    for name, g in MODES:
        W, H = W0.copy(order="F"), H0.copy(order="F")
        r = nmfx.solve(nmfx.ALSPGrad(T, maxiter=iters, tol=1e-30, gradient=g), X, W, H, track_objective=True)
        row[name] = {"objective_max_rel_err": float(rel_trace_err(r.trace, ro.trace)), "inner": int(r.info["inner_iters"]),
                     "backtracks": int(r.info["backtracks"]), "W_max_rel_err": float(np.max(np.abs(W - Wc)) / np.max(np.abs(Wc)))}
    return row


def timing(T, shape, iters, maxsubiter):
    p, n, k = shape
    X, W0, H0 = planted(p, n, k, T, seed=5)
    row = {"what": "time", "dtype": np.dtype(T).name, "shape": list(shape), "outer_iterations": iters, "maxsubiter": maxsubiter}
    with nmfx.Context(T, p, n, k) as ctx:
        ctx.set_X(X)
        for name, g in MODES:
            o = nmfx.make_opts(T, maxiter=iters, tol=1e-30, maxsubiter=maxsubiter, pg_refresh=(1 if g == "exact" else g))
            ctx.set_factors(W0, H0)
            ctx.iterate(nmfx._lib.ALG_ALSPGRAD, nmfx.make_opts(T, maxiter=1, tol=1e-30, maxsubiter=maxsubiter, pg_refresh=(1 if g == "exact" else g)))
            ctx.set_factors(W0, H0)
            t0 = time.perf_counter()
            res, _ = ctx.iterate(nmfx._lib.ALG_ALSPGRAD, o)
            dt = time.perf_counter() - t0
            row[name] = {"ms_per_outer_iteration": round(dt / iters * 1e3, 2), "inner": int(res.inner_iters), "backtracks": int(res.backtracks),
                         "objective": float(res.objvalue)}
    return row


if __name__ == "__main__":
    for T in (np.float32, np.float64):
        for shape, seed in (((40, 56, 4), 87), ((140, 300, 9), 331), ((129, 200, 100), 231), ((512, 640, 12), 77), ((1024, 768, 32), 11), ((300, 2000, 64), 3)):
            print(json.dumps(parity(T, shape, seed)), flush=True)
    print(json.dumps(timing(np.float32, (8192, 8192, 256), 3, 200)), flush=True)
    print(json.dumps(timing(np.float64, (32768, 4096, 512), 2, 200)), flush=True)      # the C5 shard shape
