"""greedy_state.py -- development aid: a realistic input of GreedyCD's W-side sweep for scripts/kbench/greedy_bench.hip.

Runs ITERS GreedyCD iterations of bench.py's C3 problem (16384^2, k = 256, f32, same synthetic data) through the library, then
forms P = H H', Z = X H', G = W P - Z with torch (data preparation only) and writes W, G (p x k, column-major) and P (k x k):
    python scripts/kbench/greedy_state.py OUT.bin [iters]
File: int64 p, k; then W, G, P as float32.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nmf.jl_amd"))
import numpy as np, torch
import nmfx, bench

out = sys.argv[1]; iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
p = n = 16384; k = 256
dev = torch.device("cuda:0")
Xt, W0, H0 = bench.synth(p, n, k, 0, n, torch.float32, dev)
with nmfx.Context(np.float32, p, n, k) as ctx:
    ctx.set_X_device(Xt.data_ptr(), p)
    ctx.set_factors(W0, H0)
    o = nmfx.make_opts(np.float32, maxiter=iters, tol=1e-30, check_every=1000)
    res, _ = ctx.iterate(5, o)
    print("greedycd", iters, "iterations, inner steps", res.inner_iters, flush=True)
    W = np.empty((p, k), np.float32, order="F"); H = np.empty((k, n), np.float32, order="F")
    ctx.get_factors(W, H)
Wd = torch.from_numpy(np.ascontiguousarray(W)).to(dev); Hd = torch.from_numpy(np.ascontiguousarray(H)).to(dev)
X = Xt.t()                                  # p x n view
P = Hd @ Hd.t(); Z = X @ Hd.t(); G = Wd @ P - Z
with open(out, "wb") as f:
    np.array([p, k], np.int64).tofile(f)
    np.asfortranarray(W).ravel(order="K").tofile(f)
    np.asfortranarray(G.cpu().numpy()).ravel(order="K").tofile(f)
    np.ascontiguousarray(P.cpu().numpy()).tofile(f)
print("wrote", out, os.path.getsize(out), "bytes")
