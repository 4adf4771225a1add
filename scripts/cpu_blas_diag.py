#!/usr/bin/env python
"""cpu_blas_diag.py -- why does bench.py's cpu_baseline (the reference's 6-GEMM MultUpdate-MSE iteration on NumPy's
OpenBLAS) run at < 100 GFLOP/s on a 256-core host?  ("next round" item 2.)

Runs in a process that has NOT imported torch (one BLAS runtime loaded), pins the OpenBLAS pool per trial, and prints ONE
JSON object: per-call seconds of every product and element-wise pass of update_wh!(::MultUpdMSE) (src/multupd.jl:98-115)
on the bench's column sample (p = 16384, ns = 2048, k = 256, f32), a square sgemm as the machine's yardstick, the host's
topology and the BLAS build.  No GPU, no oracle import: plain NumPy calls in the oracle's operand layouts.
"""
from __future__ import annotations

import json
import os
import platform
import subprocess
import sys
import time

import numpy as np

try:
    from threadpoolctl import threadpool_info, threadpool_limits
except Exception:  # noqa: BLE001
    threadpool_info = threadpool_limits = None


def best_of(f, reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def main():
    p, ns, k = 16384, 2048, 256
    if len(sys.argv) > 3:
        p, ns, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    T = np.float32
    rng = np.random.default_rng(1)
    X = np.asfortranarray(rng.random((p, ns), dtype=T))
    W = np.asfortranarray(rng.random((p, k), dtype=T))
    H = np.asfortranarray(rng.random((k, ns), dtype=T))
    WH = np.asfortranarray(W @ H)
    cap = None
    info = threadpool_info() if threadpool_info else []
    for d in info:
        if d.get("user_api") == "blas":
            cap = max(cap or 0, d.get("num_threads") or 0)
    cands = sorted({c for c in (1, 2, 4, 8, 16, 32, 64, 128, cap or 0) if c and (cap is None or c <= cap)})
    calls = {
        "WtX = W' X        (k x p)(p x ns)   :98": (lambda: W.T @ X, 2.0 * p * ns * k),
        "WtWH = W' WH      (k x p)(p x ns)   :99": (lambda: W.T @ WH, 2.0 * p * ns * k),
        "WH = W H          (p x k)(k x ns)   :104": (lambda: W @ H, 2.0 * p * ns * k),
        "XHt = X H'        (p x ns)(ns x k)  :109": (lambda: X @ H.T, 2.0 * p * ns * k),
        "WHHt = WH H'      (p x ns)(ns x k)  :110": (lambda: WH @ H.T, 2.0 * p * ns * k),
        "WH = W H (2nd)                       :115": (lambda: W @ H, 2.0 * p * ns * k),
    }
    WtX = W.T @ X
    WtWH = W.T @ WH
    XHt = X @ H.T
    WHHt = WH @ H.T
    elementwise = {
        "H .*= max(0, WtX - l) ./ (WtWH + d)  :101-103": (lambda: np.multiply(H, np.maximum(T(0), WtX - T(0)) / (WtWH + T(1e-3)), out=np.empty_like(H)), 5.0 * k * ns * 4),
        "W .*= max(0, XHt - l) ./ (WHHt + d)  :112-114": (lambda: np.multiply(W, np.maximum(T(0), XHt - T(0)) / (WHHt + T(1e-3)), out=np.empty_like(W)), 5.0 * p * k * 4),
        "preW, preH copies                    common.jl:66-67": (lambda: (W.copy(), H.copy()), 2.0 * (p * k + k * ns) * 4),
    }
    sq = 4096
    A = rng.random((sq, sq), dtype=T)
    B = rng.random((sq, sq), dtype=T)
    rows = []
    for c in cands:
        cm = threadpool_limits(limits=c, user_api="blas") if threadpool_limits else None
        try:
            (W.T @ X)   # warm the pool at this size
            r = {"blas_threads": c, "calls": {}, "gemm_seconds": 0.0}
            for name, (f, fl) in calls.items():
                t = best_of(f)
                r["calls"][name] = {"s": round(t, 4), "gflops": round(fl / t / 1e9, 1)}
                r["gemm_seconds"] += t
            r["gemm_seconds"] = round(r["gemm_seconds"], 4)
            r["gemm_gflops"] = round(12.0 * p * ns * k / r["gemm_seconds"] / 1e9, 1)
            t = best_of(lambda: A @ B, 2)
            r["sgemm_4096_cubed_gflops"] = round(2.0 * sq ** 3 / t / 1e9, 1)
            rows.append(r)
        finally:
            if cm is not None:
                cm.restore_original_limits()
    ew = {}
    for name, (f, by) in elementwise.items():
        t = best_of(f)
        ew[name] = {"s": round(t, 4), "gbs": round(by / t / 1e9, 2)}
    try:
        lscpu = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        keep = ("Model name", "Socket(s)", "Core(s) per socket", "Thread(s) per core", "CPU(s):", "NUMA node(s)", "L3 cache", "CPU max MHz")
        lscpu = {ln.split(":")[0].strip(): ln.split(":", 1)[1].strip() for ln in lscpu.splitlines() if ln.split(":")[0].strip() in
                 [k_.rstrip(":") for k_ in keep]}
    except Exception as e:  # noqa: BLE001
        lscpu = {"error": repr(e)}
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:  # noqa: BLE001
        aff = None
    out = {"shape": {"p": p, "ns": ns, "k": k, "dtype": "f32"}, "host": {"cpu_count": os.cpu_count(), "affinity": aff, "lscpu": lscpu,
                                                                            "machine": platform.machine()},
           "blas": [{"api": d.get("internal_api"), "version": d.get("version"), "threads": d.get("num_threads"), "arch": d.get("architecture"),
                     "lib": os.path.basename(d.get("filepath") or "")} for d in info],
           "thread_trials": rows, "elementwise_single_thread": ew,
           "env": {k_: os.environ.get(k_) for k_ in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_CORETYPE")}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
