"""One rank of a multi-PROCESS run of the sharded path over the peer-to-peer transport (csrc/peer.hpp), launched by
tests/test_gpu_peer.py (and usable by hand):  several processes, all on device NMFX_WORKER_DEVICE (default 0) -- the 1-GPU box's
stand-in for one process per GPU -- exchange their window handles through a gloo rendezvous, map each other's windows with
hipIpcOpenMemHandle and solve the column-sharded problem.  Writes W, H shard, objective trace and counters to --out."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("nmf.jl_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, sub))

import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import torch.distributed as dist  # noqa: E402

import nmfx  # noqa: E402
from problems import planted  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--alg", default="multmse")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--p", type=int, default=300)
    ap.add_argument("--n", type=int, default=530)
    ap.add_argument("--k", type=int, default=6)
    ap.add_argument("--seed", type=int, default=17)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--lam", type=float, default=0.0)
    ap.add_argument("--mode", default="row_sharded")
    ap.add_argument("--update-h", type=int, default=1)
    ap.add_argument("--track", type=int, default=1)
    ap.add_argument("--tol", type=float, default=1e-30)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    T = np.float32 if a.dtype == "f32" else np.float64
    L = nmfx._lib
    algid = {"multmse": L.ALG_MULTMSE, "multdiv": L.ALG_MULTDIV, "projals": L.ALG_PROJALS, "alspgrad": L.ALG_ALSPGRAD, "cd": L.ALG_CD,
             "greedycd": L.ALG_GREEDYCD}[a.alg]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{a.port}", rank=a.rank, world_size=a.world)
    X, W0, H0 = planted(a.p, a.n, a.k, T, seed=a.seed, normalize=(a.alg != "projals"))
    c0, c1 = nmfx.dist.shard_range(a.n, a.rank, a.world)
    dev = int(os.environ.get("NMFX_WORKER_DEVICE", "0"))
    with nmfx.Context(T, a.p, c1 - c0, a.k, device=dev) as ctx:
        nmfx.dist.init_comm(ctx, transport="p2p_only")             # before set_X: the row padding may change
        ctx.comm_set_mode(a.mode)
        ctx.set_X(np.asfortranarray(X[:, c0:c1]))
        W, H = W0.copy(order="F"), np.asfortranarray(H0[:, c0:c1].copy())
        kw = dict(maxiter=a.iters, tol=a.tol, lambda_w=a.lam, lambda_h=a.lam, track_objective=bool(a.track), update_H=bool(a.update_h))
        res, trace = ctx.solve(algid, nmfx.make_opts(T, **kw), W, H)
        served, based = ctx.comm_p2p_stats()
    np.savez(a.out, W=W, H=H, trace=(np.asarray(trace) if trace is not None else np.zeros(0)), niters=res.niters, converged=int(res.converged),
             inner=res.inner_iters, backtracks=res.backtracks, objvalue=res.objvalue, c0=c0, c1=c1, served=served, based=based)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
