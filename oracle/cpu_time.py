#!/usr/bin/env python
"""cpu_time.py -- TEST / MEASUREMENT INFRASTRUCTURE (bench.py's `cpu_baseline` leg), never product code.

Times NMF.jl's CPU path for MultUpdate-MSE -- the oracle's restatement of the reference's operation sequence with the
state of `prepare_state` allocated ONCE (nmf_oracle._MultMSEState: src/multupd.jl:63-80 state, :83-116 update_wh!,
src/common.jl:66-73 preW/preH copies + stop_condition) -- in a process of its own, so that exactly ONE BLAS runtime is
loaded (bench.py's own process carries torch's OpenMP pool and a second OpenBLAS beside NumPy's).

    python oracle/cpu_time.py sample.npz            # X (p x ns), W0 (p x k), H0 (k x ns)

Prints ONE JSON object: for every BLAS pool size tried the seconds per call site (each mul!, each element-wise loop,
the copies, stop_condition's two passes), the GEMM phases' GFLOP/s on their own, and the fastest setting.  Round 3 timed the
ALLOCATING form of the oracle inside bench.py's process: 1.07 s per sample iteration, of which ~1.0 s were first-touch
page faults of three fresh p x ns temporaries per iteration (the GPU box is a micro-VM: 0.25 GB/s on fresh pages) --
work NMF.jl does not do.  The GEMM phases alone were always fast (scripts/cpu_blas_diag.py).
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nmf_oracle as orc  # noqa: E402

try:
    from threadpoolctl import threadpool_info, threadpool_limits
except Exception:  # noqa: BLE001
    threadpool_info = threadpool_limits = None


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("sample")                                      # .npz: X (p x ns), W0 (p x k), H0 (k x ns); with --x-npy: W0, H0 only
    ap.add_argument("budget_s", nargs="?", type=float, default=4.0)
    ap.add_argument("n_full", nargs="?", type=int, default=None)   # columns of the FULL problem the sample stands for
    ap.add_argument("--x-npy", default=None, help="X as an uncompressed .npy (the FULL problem: too large for an .npz round trip)")
    ap.add_argument("--threads", default=None, help="comma-separated BLAS pool sizes to try (default: the pool's cap and 1/2 ... 1/16 of it)")
    ap.add_argument("--min-iters", type=int, default=1, help="timed iterations per setting, at least (the budget only stops the loop beyond it)")
    ap.add_argument("--max-iters", type=int, default=6)
    a = ap.parse_args()
    d = np.load(a.sample)
    budget_s, n_full = a.budget_s, a.n_full
    if a.x_npy:
        X = np.load(a.x_npy, mmap_mode="r")
        X = np.array(X, order="F")             # a private copy: this process's own pages, touched here and not in the timed loop
        W0, H0 = (np.asfortranarray(d[k]) for k in ("W0", "H0"))
    else:
        X, W0, H0 = (np.asfortranarray(d[k]) for k in ("X", "W0", "H0"))
    T = X.dtype.type
    p, ns = X.shape
    k = W0.shape[1]
    tiny = float(np.finfo(T).tiny)
    o = orc.resolve_opts(orc.MULTMSE, T, orc.Opts(maxiter=1, tol=tiny))
    info = threadpool_info() if threadpool_info else []
    blas = [x for x in info if x.get("user_api") == "blas"]
    cap = max((x.get("num_threads") or 0) for x in blas) if blas else None
    cands = [None]
    if cap and threadpool_limits:
        cands = sorted({c for c in (cap, cap // 2, cap // 4, cap // 8, cap // 16) if c >= 1}, reverse=True)
        if a.threads:
            cands = [min(cap, max(1, int(c))) for c in a.threads.split(",")]
    trials = []
    for c in cands:
        cm = threadpool_limits(limits=c, user_api="blas") if (c and threadpool_limits) else None
        try:
            Ws, Hs = W0.copy(order="F"), H0.copy(order="F")
            st = orc._MultMSEState(T, o, X, Ws, Hs)               # prepare_state: NOT timed
            st.update(X, Ws, Hs)                                   # warm-up (BLAS threads, first touch of the state)
            phases, iters, used = {}, 0, 0.0
            while iters < a.max_iters and (used < budget_s or iters < a.min_iters):
                t0 = time.perf_counter()
                np.copyto(st.preW, Ws)                            # common.jl:66
                t1 = time.perf_counter()
                np.copyto(st.preH, Hs)                            # common.jl:67
                t2 = time.perf_counter()
                st.update(X, Ws, Hs, phases)                      # common.jl:70
                st.stop_condition(Ws, Hs, tiny, sequential=False, phases=phases)   # common.jl:73 (np.sum form: see nmf_oracle.py)
                t3 = time.perf_counter()
                phases["copyto! preW           common.jl:66"] = phases.get("copyto! preW           common.jl:66", 0.0) + (t1 - t0)
                phases["copyto! preH           common.jl:67"] = phases.get("copyto! preH           common.jl:67", 0.0) + (t2 - t1)
                used += t3 - t0
                iters += 1
            per = {n_: round(v / iters, 5) for n_, v in phases.items()}
            gemm_s = sum(v for n_, v in per.items() if n_.startswith("mul!"))
            # what scales with the number of columns (every mul!, everything that touches H) and what does not (the p x k passes over W)
            fixed_s = sum(v for n_, v in per.items() if n_.startswith(("W loop", "copyto! preW", "stop_condition W")))
            tr = {"blas_threads": c, "iters": iters, "seconds_per_sample_iter": round(used / iters, 5), "phase_seconds": per,
                  "gemm_seconds": round(gemm_s, 5), "gemm_gflops": round(12.0 * p * ns * k / gemm_s / 1e9, 1),
                  "non_gemm_seconds": round(used / iters - gemm_s, 5), "column_independent_seconds": round(fixed_s, 5)}
            if n_full:
                tr["seconds_per_iter_full_est"] = round((used / iters - fixed_s) * (n_full / ns) + fixed_s, 5)
            trials.append(tr)
        finally:
            if cm is not None:
                cm.restore_original_limits()
    best = min(trials, key=lambda t: t["seconds_per_sample_iter"])
    print(json.dumps({"p": p, "ns": ns, "k": k, "dtype": np.dtype(T).name, "host_cores": os.cpu_count(), "pool_cap": cap, "best": best, "thread_trials": trials,
                      "blas": [{"api": x.get("internal_api"), "version": x.get("version"), "threads": x.get("num_threads"), "arch": x.get("architecture"),
                                "lib": os.path.basename(x.get("filepath") or "")} for x in info]}))


if __name__ == "__main__":
    main()
