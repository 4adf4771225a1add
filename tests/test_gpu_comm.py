"""RCCL path on the 1-GPU box.

A communicator of ONE rank normally short-cuts to the single-GPU code, so nothing of the exchange step runs.  With
NMFX_FORCE_SHARDED=1 (read when the context is created, csrc/solver.hpp) the sharded code path is kept for nranks == 1: the
blocked combine, ncclReduceScatter + grouped ncclAllReduce, the rank's row block of the W update, ncclAllGather (byte chunks),
the objective / line-search scalar all-reduces and the pipelined exchange on its second stream ALL execute under RCCL, as
identities.  The same forced path is run on the in-process transport (LocalComm, one rank): the two transports must agree bit
for bit, and both must reproduce the plain single-GPU run."""
import os

import numpy as np
import pytest

import nmfx
from problems import planted, rel_trace_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("alg_name", ["multmse", "multdiv", "projals", "alspgrad"])
def test_nranks1_comm_is_identity(built, alg_name):
    T = np.float32
    p, n, k = 260, 300, 6
    X, W0, H0 = planted(p, n, k, T, seed=5, normalize=(alg_name != "projals"))
    if alg_name in ("multmse", "multdiv"):
        alg = nmfx.MultUpdate(T, obj=alg_name[4:], maxiter=12, tol=1e-30)
    elif alg_name == "projals":
        alg = nmfx.ProjectedALS(T, maxiter=8, tol=1e-30, lambda_w=0.05, lambda_h=0.05)
    else:
        alg = nmfx.ALSPGrad(T, maxiter=4, tol=1e-30)
    outs = []
    for use_comm in (False, True):
        W, H = W0.copy(order="F"), H0.copy(order="F")
        with nmfx.Context(T, p, n, k) as ctx:
            ctx.set_X(X)
            if use_comm:
                ctx.comm_init(nmfx.comm_unique_id(), 0, 1)
            r = nmfx.solve(alg, X, W, H, ctx=ctx, track_objective=True)
        outs.append((W, H, r))
    (W0_, H0_, r0), (W1, H1, r1) = outs
    assert r0.niters == r1.niters
    assert np.array_equal(r0.trace, r1.trace)
    assert np.array_equal(W0_, W1) and np.array_equal(H0_, H1)


L = nmfx._lib
ALGS = {"multmse": L.ALG_MULTMSE, "multdiv": L.ALG_MULTDIV, "projals": L.ALG_PROJALS, "alspgrad": L.ALG_ALSPGRAD,
        "cd": L.ALG_CD, "greedycd": L.ALG_GREEDYCD}


def _forced_run(T, X, W0, H0, alg, kw, transport, mode):
    p, n = X.shape
    k = W0.shape[1]
    os.environ["NMFX_FORCE_SHARDED"] = "1"
    group = None
    try:
        with nmfx.Context(T, p, n, k) as ctx:
            if transport == "rccl":
                ctx.comm_init(nmfx.comm_unique_id(), 0, 1)
            elif transport == "local":
                group = nmfx.LocalGroup(1)
                ctx.comm_init_local(group, 0)
            if transport != "none":
                ctx.comm_set_mode(mode)
            ctx.set_X(X)
            W, H = W0.copy(order="F"), H0.copy(order="F")
            res, trace = ctx.solve(ALGS[alg], nmfx.make_opts(T, **kw), W, H)
    finally:
        os.environ.pop("NMFX_FORCE_SHARDED", None)
        if group is not None:
            group.close()
    return W, H, res, trace


@pytest.mark.parametrize("alg,mode", [("multmse", "row_sharded"), ("multmse", "pipelined"), ("multmse", "replicated_w"),
                                      ("multdiv", "row_sharded"), ("projals", "row_sharded"), ("alspgrad", "row_sharded"),
                                      ("cd", "row_sharded"), ("greedycd", "row_sharded"), ("projals", "replicated_w")])
@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_forced_sharded_path_under_rccl_is_identity(built, alg, mode, T):
    """ncclReduceScatter / ncclAllGather / the grouped all-reduces of the DEFAULT multi-GPU mode (and of the pipelined and
    replicated modes) executed by RCCL with one rank: bit-identical to the in-process transport running the same sharded code,
    and the same trajectory as the plain single-GPU run."""
    p, n, k = 300, 530, (200 if mode == "pipelined" else 6)       # pipelined: K = 256 (the fused-Gram launches it is built on)
    X, W0, H0 = planted(p, n, k, T, seed=11, normalize=(alg != "projals"))
    lam = {"projals": 0.05, "multdiv": float(np.sqrt(np.finfo(T).eps)), "multmse": 1e-4}.get(alg, 0.0)
    iters = 4 if alg == "alspgrad" else 8
    kw = dict(maxiter=iters, tol=1e-30, lambda_w=lam, lambda_h=lam, track_objective=True)
    Wr, Hr, rr, tr = _forced_run(T, X, W0, H0, alg, kw, "rccl", mode)
    Wl, Hl, rl, tl = _forced_run(T, X, W0, H0, alg, kw, "local", mode)
    assert rr.niters == rl.niters == iters
    assert np.array_equal(tr, tl) and np.array_equal(Wr, Wl) and np.array_equal(Hr, Hl)        # transport = identity
    if alg == "alspgrad":
        assert rr.inner_iters == rl.inner_iters and rr.backtracks == rl.backtracks
    W1, H1, r1, t1 = _forced_run(T, X, W0, H0, alg, kw, "none", mode)
    tol = {np.float64: 1e-9, np.float32: 2e-5}[T]
    if T == np.float32 and alg == "projals":
        tol = 2e-3
    if T == np.float32 and alg == "greedycd":
        tol = 2e-2
    assert rel_trace_err(tr, t1) < tol
    assert np.max(np.abs(Wr - W1)) <= 100 * tol * np.max(np.abs(W1))
    assert np.max(np.abs(Hr - H1)) <= 100 * tol * np.max(np.abs(H1))


def test_forced_sharded_untracked_pipelined_and_stop_rule_under_rccl(built):
    """The pipelined mode with W left in flight between iterations (no objective tracking) and a tolerance that stops the
    solve: RCCL collectives on the second stream, deferred stop check -- same niters / objective as the plain run."""
    T = np.float64
    p, n, k = 256, 384, 130
    X, W0, H0 = planted(p, n, k, T, seed=3, k0=4)
    kw = dict(maxiter=300, tol=3e-3, lambda_w=0.0, lambda_h=0.0, check_every=4)
    W1, H1, r1, _ = _forced_run(T, X, W0, H0, "multmse", kw, "none", "row_sharded")
    assert r1.converged and 3 < r1.niters < 300
    for mode in ("pipelined", "row_sharded"):
        Wr, Hr, rr, _ = _forced_run(T, X, W0, H0, "multmse", kw, "rccl", mode)
        assert rr.converged and rr.niters == r1.niters
        assert abs(rr.objvalue - r1.objvalue) <= 1e-9 * abs(r1.objvalue)
        assert np.max(np.abs(Wr - W1)) <= 1e-7 * np.max(np.abs(W1))


def test_leading_dimension_limit(built):
    """The fused epilogues address a wave tile with 32-bit byte offsets: 256 * ld * sizeof(T) must stay below 2^32
     -- a larger p is refused up front with NMFX_ERR_UNSUPPORTED, before anything is allocated."""
    with pytest.raises(nmfx.NMFXError, match="too large"):
        nmfx.Context(np.float32, 4_194_304, 2, 2)
    with pytest.raises(nmfx.NMFXError, match="too large"):
        nmfx.Context(np.float64, 2_097_152, 2, 2)
