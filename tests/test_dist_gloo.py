"""N>1 path on CPU: world_size=2 over gloo.

What this pins without a GPU: (1) both sharded formulations reproduce the unsharded reference trajectory (oracle),
including ragged shards: the replicated-W form (ONE packed all-reduce per outer iteration) for multmse / multdiv / projals /
cd / greedycd (the last with its max all-reduce of p_init on the sharded H side), and the ROW-SHARDED W side (reduce-scatter
of X_g H_g' by row blocks, each rank updates its rows, all-gather) for multmse / multdiv / projals and for alspgrad with its
all-reduced line-search scalars (inner-iteration and back-track counters equal the unsharded oracle's); (2) the host plumbing
the GPU path uses (shard_range, packed layout, unique-id broadcast) under a real process group.
libnmfx's own sharded code runs with 2-8 ranks on the GPU box through the in-process transport
(tests/test_gpu_localcomm.py); the RCCL transport with nranks = 1 there (tests/test_gpu_comm.py) and with 8 ranks by the
driver's scaling bench."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import nmf_oracle as orc
import nmfx
from problems import planted


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, alg, T_name, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for sub in ("../nmf.jl_amd", "../oracle", "."):
        sys.path.insert(0, os.path.join(here, sub))
    import sharded_model as sm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T = np.dtype(T_name).type
    rows = alg.endswith("+rows")
    alg = alg.split("+")[0]
    p, n, k = 37, 53, 4                                # 53 columns over 2 ranks: ragged (27 + 26)
    X, W0, H0 = planted(p, n, k, T, seed=11, normalize=(alg != "projals"))
    c0, c1 = nmfx.dist.shard_range(n, rank, world)
    Xg = np.asfortranarray(X[:, c0:c1])
    W, Hg = W0.copy(order="F"), np.asfortranarray(H0[:, c0:c1].copy())

    def allreduce(a):
        t = torch.from_numpy(np.ascontiguousarray(a))
        dist.all_reduce(t)
        return t.numpy()

    # unique-id broadcast used by nmfx.dist.init_comm (payload is opaque bytes)
    uid = nmfx.dist.broadcast_unique_id(lambda: bytes(range(128)))
    assert uid == bytes(range(128))
    def allmax(v):
        t = torch.tensor([float(v)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return T(t.item())

    _allreduce = allreduce

    class Comm:
        """reduce-scatter / all-gather of row blocks over gloo (gloo has no reduce_scatter: one reduce per destination)."""
        allreduce = staticmethod(_allreduce)

        @staticmethod
        def row_range(pp):
            return sm.row_block(pp, rank, world)

        @staticmethod
        def reduce_scatter_rows(A):
            mine = None
            for dst in range(world):
                r0, r1 = sm.row_block(A.shape[0], dst, world)
                t = torch.from_numpy(np.ascontiguousarray(A[r0:r1]))
                dist.reduce(t, dst=dst)
                if dst == rank:
                    mine = t.numpy().copy()
            return mine

        @staticmethod
        def all_gather_rows(Ar):
            parts = [None] * world
            dist.all_gather_object(parts, Ar)
            return np.concatenate(parts, axis=0)

    lam = 0.05 if alg == "projals" else (0.0 if alg in ("cd", "greedycd", "alspgrad") else 1e-4)
    o = orc.resolve_opts(orc.ALG_NAMES[alg], T, orc.Opts(lambda_w=lam, lambda_h=lam))
    oalg = "multmse" if alg == "alspgrad" else alg                    # objective: 0.5 * sqL2dist for both
    trace = [sm.objective(oalg, Xg, W, Hg, allreduce)]
    state = {"tolg": T(o.tolg), "inner": 0, "backtracks": 0}
    for _ in range(6):
        if alg == "alspgrad":
            sm.step_alspgrad(Xg, W, Hg, state, Comm)
        elif alg in ("cd", "greedycd"):
            sm.step_cd(alg, Xg, W, Hg, o.lambda_w, o.lambda_h, allreduce, allmax)
        elif rows:
            sm.step_rows(alg, Xg, W, Hg, o.lambda_w, o.lambda_h, o.delta, Comm)
        else:
            sm.step(alg, Xg, W, Hg, o.lambda_w, o.lambda_h, o.delta, allreduce)
        trace.append(sm.objective(oalg, Xg, W, Hg, allreduce))
    Hs = [None] * world
    dist.all_gather_object(Hs, (c0, c1, Hg))
    Ws = [None] * world
    dist.all_gather_object(Ws, W)
    assert all(np.array_equal(Ws[0], w) for w in Ws)                 # W is identical on every rank after the exchange
    if rank == 0:
        q.put((trace, W, Hs, (state["inner"], state["backtracks"])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("alg", ["multmse", "multdiv", "projals", "cd", "greedycd",
                                 "multmse+rows", "multdiv+rows", "projals+rows", "alspgrad+rows"])
def test_sharded_formulation_matches_unsharded_reference(alg):
    T = np.float64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, alg, "float64", q)) for r in range(2)]
    for pr in procs:
        pr.start()
    trace, W, Hs, counters = q.get(timeout=120)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    alg = alg.split("+")[0]
    X, W0, H0 = planted(37, 53, 4, T, seed=11, normalize=(alg != "projals"))
    lam = 0.05 if alg == "projals" else (0.0 if alg in ("cd", "greedycd", "alspgrad") else 1e-4)
    Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
    ro = orc.solve(alg, X, Wc, Hc, orc.Opts(maxiter=6, tol=1e-30, lambda_w=lam, lambda_h=lam, track_objective=True))
    ref = np.array(ro.trace)
    if alg == "projals":                                # oracle adds the regularisers to the objective; the model does not
        ref_last = 0.5 * np.sum((X - Wc @ Hc) ** 2)
        assert abs(trace[-1] - ref_last) <= 1e-8 * ref_last
    else:
        assert np.max(np.abs(np.array(trace) - ref) / ref) < 1e-10
    H = np.zeros_like(Hc)
    for c0, c1, Hg in Hs:
        H[:, c0:c1] = Hg
    assert np.max(np.abs(W - Wc)) <= 1e-8 * np.max(np.abs(Wc))
    assert np.max(np.abs(H - Hc)) <= 1e-8 * np.max(np.abs(Hc))
    if alg == "alspgrad":      # one global step size per back-track: the sharded run takes exactly the unsharded run's steps
        assert counters == (ro.counters["inner"], ro.counters["backtracks"])


def test_shard_range_partitions_columns():
    for n, world in ((16384, 8), (53, 2), (10, 3), (7, 7), (131072, 8)):
        cuts = [nmfx.dist.shard_range(n, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        for a, b in zip(cuts, cuts[1:]):
            assert a[1] == b[0]
        sizes = [c1 - c0 for c0, c1 in cuts]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        nmfx.dist.shard_range(3, 0, 4)


def test_packed_layout():
    lay = nmfx.dist.packed_layout(512, 128)
    assert lay["XHt"] == (0, 512 * 128) and lay["HHt"][1] - lay["HHt"][0] == 128 * 128
    assert lay["sH"][1] == lay["count"] == 512 * 128 + 128 * 128 + 128
