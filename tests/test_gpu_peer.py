"""The peer-to-peer transport of the sharded path (csrc/peer.hpp, include/nmfx.h nmfx_comm_init_p2p / _p2p_export / _p2p_attach):
every rank pushes its contributions straight into the peers' windows, device-side flags order producer and consumer, the consumer
adds the ranks' contributions in rank order.  That order is the in-process group's (csrc/comm.hpp, LocalComm), so the two
transports must agree BIT FOR BIT -- factors, objective trajectory, ALSPGrad's counters -- for all six algorithms:
  * in one process (contexts on threads, windows shared by pointer), 2 and 4 ranks, both element types;
  * across PROCESSES on device 0 (windows mapped with hipIpcOpenMemHandle, handles shipped through a gloo rendezvous), 2, 4 and 8
    ranks -- the one-process-per-GPU layout of bench.py with all ranks on the one GPU the test box has.
The reference has no distributed path."""
import os
import socket
import subprocess
import sys
import tempfile
import threading

import numpy as np
import pytest

import nmfx
from problems import planted
from test_gpu_localcomm import ALG, lam_for, run_sharded

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_peer_threads(T, X, W0, H0, alg, opts_kw, G, mode="row_sharded", timeout=300, wrap_local=False):
    """G in-process ranks over the peer windows (wrap_local: an in-process group underneath as the fallback transport)."""
    p, n = X.shape
    k = W0.shape[1]
    out, errs, handles = [None] * G, [], [None] * G
    bar = threading.Barrier(G)
    group = nmfx.LocalGroup(G) if wrap_local else None

    def worker(r):
        try:
            c0, c1 = nmfx.dist.shard_range(n, r, G)
            with nmfx.Context(T, p, c1 - c0, k) as ctx:
                if wrap_local:
                    ctx.comm_init_local(group, r)
                else:
                    ctx.comm_init_p2p(r, G)
                handles[r] = ctx.comm_p2p_export()
                bar.wait(timeout)
                ctx.comm_p2p_attach(handles)
                ctx.comm_set_mode(mode)
                ctx.set_X(np.asfortranarray(X[:, c0:c1]))
                W, H = W0.copy(order="F"), np.asfortranarray(H0[:, c0:c1].copy())
                res, trace = ctx.solve(ALG[alg], nmfx.make_opts(T, **opts_kw), W, H)
                out[r] = (c0, c1, W, H, res, trace, ctx.comm_p2p_stats())
                bar.wait(timeout)                                 # nobody frees a window a peer may still be storing into
        except Exception as e:  # noqa: BLE001
            errs.append((r, repr(e)))
            bar.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(G)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout)
    assert not errs, errs
    assert all(o is not None for o in out), "a rank did not finish"
    if group is not None:
        group.close()
    H = np.zeros_like(H0)
    for c0, c1, _, Hg, *_ in out:
        H[:, c0:c1] = Hg
    return out[0][2], H, [(o[4], o[5]) for o in out], [o[2] for o in out], [o[6] for o in out]


def same_run(a, b, alg):
    (Wa, Ha, ra, Walla), (Wb, Hb, rb, Wallb) = a, b
    assert np.array_equal(Wa, Wb) and np.array_equal(Ha, Hb)
    for W in list(Walla) + list(Wallb):
        assert np.array_equal(W, Wa)
    for (r1, t1), (r2, t2) in zip(ra, rb):
        assert r1.niters == r2.niters and r1.converged == r2.converged
        assert (t1 is None and t2 is None) or np.array_equal(np.asarray(t1, dtype=np.float64), np.asarray(t2, dtype=np.float64), equal_nan=True)
        assert r1.objvalue == r2.objvalue or (np.isnan(r1.objvalue) and np.isnan(r2.objvalue))
        if alg == "alspgrad":
            assert r1.inner_iters == r2.inner_iters and r1.backtracks == r2.backtracks


@pytest.mark.parametrize("alg", ["multmse", "multdiv", "projals", "alspgrad", "cd", "greedycd"])
@pytest.mark.parametrize("G", [2, 4])
@pytest.mark.parametrize("T", [np.float64, np.float32])
def test_peer_windows_equal_the_in_process_group_bit_for_bit(built, alg, G, T):
    p, n, k = 300, 530, 6                                           # ragged column shards; p padded to 128*G rows
    X, W0, H0 = planted(p, n, k, T, seed=17, normalize=(alg != "projals"))
    lam = lam_for(alg, T)
    kw = dict(maxiter=5 if alg == "alspgrad" else 10, tol=1e-30, lambda_w=lam, lambda_h=lam, track_objective=True)
    ref = run_sharded(T, X, W0, H0, alg, kw, G)
    *got, stats = run_peer_threads(T, X, W0, H0, alg, kw, G)
    same_run(ref, tuple(got), alg)
    assert all(s[0] > 0 and s[1] == 0 for s in stats), stats      # every collective was served by the windows


@pytest.mark.parametrize("alg,mode", [("multmse", "replicated_w"), ("projals", "replicated_w"), ("multmse", "pipelined")])
def test_peer_other_modes(built, alg, mode):
    """replicated_w: the packed all-reduce is larger than a slot and travels in slot-sized pieces.  pipelined: its collectives run
    on a second stream -- the windows cannot order those, the wrapped in-process group serves them."""
    T = np.float64
    # (the pipelined exchange exists for K % 128 == 0 and whole 128-row tiles per rank and super-chunk: p = 512, k = 100 on 2 ranks;
    # smaller shapes run the plain row-sharded step whatever the mode says)
    p, n, k = (512, 600, 100) if mode == "pipelined" else (260, 410, 5)
    X, W0, H0 = planted(p, n, k, T, seed=23, normalize=(alg != "projals"))
    lam = lam_for(alg, T)
    kw = dict(maxiter=6, tol=1e-30, lambda_w=lam, lambda_h=lam, track_objective=True)
    ref = run_sharded(T, X, W0, H0, alg, kw, 2, mode=mode)
    *got, stats = run_peer_threads(T, X, W0, H0, alg, kw, 2, mode=mode, wrap_local=(mode == "pipelined"))
    same_run(ref, tuple(got), alg)
    if mode == "pipelined":
        # the windows serve the main stream only (one sequence counter, two parities: safe on ONE stream); the reduce-scatters and
        # all-gathers of the second stream are handed to the wrapped transport -- both counters must have moved
        assert all(s[0] > 0 and s[1] > 0 for s in stats), stats


def test_peer_stop_rule_and_update_h_false(built):
    T = np.float64
    p, n, k = 256, 384, 4
    X, W0, H0 = planted(p, n, k, T, seed=3)
    for kw in (dict(maxiter=400, tol=1e-3, lambda_w=0.0, lambda_h=0.0), dict(maxiter=8, tol=1e-30, update_H=False, track_objective=True)):
        ref = run_sharded(T, X, W0, H0, "multmse", kw, 2)
        *got, _ = run_peer_threads(T, X, W0, H0, "multmse", kw, 2)
        same_run(ref, tuple(got), "multmse")
    assert ref[2][0][0].niters == 8


def test_peer_ragged_shards_straddling_a_padding_boundary(built):
    """n = 513 on 2 ranks: the shards pad to 512 and 256 columns, i.e. the ranks' GEMM grids differ (the
    line-search scalars must not be all-reduced with a rank-dependent count).  Both transports, ALSPGrad, counters equal."""
    T = np.float64
    p, n, k = 200, 513, 5
    X, W0, H0 = planted(p, n, k, T, seed=29)
    kw = dict(maxiter=4, tol=1e-30, track_objective=True)
    ref = run_sharded(T, X, W0, H0, "alspgrad", kw, 2)
    *got, _ = run_peer_threads(T, X, W0, H0, "alspgrad", kw, 2)
    same_run(ref, tuple(got), "alspgrad")
    import nmf_oracle as orc
    Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
    ro = orc.solve("alspgrad", X, Wc, Hc, orc.Opts(maxiter=4, tol=1e-30, track_objective=True))
    assert ref[2][0][0].inner_iters == ro.counters["inner"] and ref[2][0][0].backtracks == ro.counters["backtracks"]
    assert np.max(np.abs(ref[0] - Wc)) <= 1e-6 * np.max(np.abs(Wc))


@pytest.mark.parametrize("kw_extra", [dict(), dict(update_H=False), dict(track_objective=False)])
def test_fused_step_with_the_product_stored_straight_into_the_peers_slots(built, kw_extra, monkeypatch):
    """The fused row-sharded MultUpdate-MSE step at a shape where X_g H_g' is stored UNSPLIT (local contraction 1024 .. 4096 columns,
    >= one tile per CU: p = 16384, k = 256, 4 ranks x 1100 columns): on the peer transport the product's epilogue writes row block g
    straight into rank g's receive slot (EpiStorePeer).  Bit-identical to the in-process group (whose product stores into the blocked
    send buffer), and equal to the UNFUSED sequence (NMFX_RS_FUSED=0: W'W over all rows instead of the ranks' own-rows Grams) to
    rounding -- ADVICE round 3: the direct-store branch and the own-rows W'W all-reduce had only ever run with one rank."""
    T = np.float32
    p, n, k, G = 16384, 4400, 256, 4
    X, W0, H0 = planted(p, n, k, T, seed=41, k0=24)
    kw = dict(maxiter=4, tol=1e-30, lambda_w=1e-4, lambda_h=1e-4, track_objective=True)
    kw.update(kw_extra)
    ref = run_sharded(T, X, W0, H0, "multmse", kw, G)
    *got, stats = run_peer_threads(T, X, W0, H0, "multmse", kw, G)
    same_run(ref, tuple(got), "multmse")
    monkeypatch.setenv("NMFX_RS_FUSED", "0")
    unf = run_sharded(T, X, W0, H0, "multmse", kw, G)
    assert np.max(np.abs(unf[0] - ref[0])) <= 2e-5 * np.max(np.abs(ref[0])) and np.max(np.abs(unf[1] - ref[1])) <= 2e-5 * np.max(np.abs(ref[1]))
    assert abs(unf[2][0][0].objvalue - ref[2][0][0].objvalue) <= 1e-5 * abs(ref[2][0][0].objvalue)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_peer_processes(T, shape, alg, G, iters, lam, seed=17, mode="row_sharded", timeout=600, extra=()):
    p, n, k = shape
    port = free_port()
    env = dict(os.environ, NMFX_P2P_TIMEOUT_S="120", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    with tempfile.TemporaryDirectory() as td:
        procs = []
        for r in range(G):
            cmd = [sys.executable, os.path.join(ROOT, "tests", "peer_worker.py"), "--rank", str(r), "--world", str(G), "--port", str(port), "--alg", alg,
                   "--dtype", "f32" if T == np.float32 else "f64", "--p", str(p), "--n", str(n), "--k", str(k), "--seed", str(seed), "--iters", str(iters),
                   "--lam", repr(lam), "--mode", mode, "--out", os.path.join(td, f"r{r}.npz"), *extra]
            procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        outs = []
        for pr in procs:
            try:
                o, _ = pr.communicate(timeout=timeout)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
            outs.append(o)
        assert all(pr.returncode == 0 for pr in procs), "\n".join(o[-1500:] for o in outs)
        res = [dict(np.load(os.path.join(td, f"r{r}.npz"))) for r in range(G)]
    return res


@pytest.mark.parametrize("alg,T,G,shape,iters", [
    ("multmse", np.float32, 2, (300, 530, 6), 10),
    ("multmse", np.float32, 4, (1024, 1100, 256), 6),      # K = 256, whole 128-row tiles per rank: the fused row-sharded step
    ("multmse", np.float32, 8, (2048, 2300, 256), 6),      # the 8-rank layout of the driver's run at 1/8 scale
    ("alspgrad", np.float64, 4, (300, 530, 6), 4),         # the line-search scalars travel inside the decision kernels
    ("alspgrad", np.float64, 8, (300, 513, 6), 2),        # (8 processes time-sharing one GPU: ~30 us per in-kernel all-reduce, ~25 s per outer iteration)
    ("projals", np.float64, 2, (300, 530, 6), 8),
    ("multdiv", np.float32, 4, (300, 530, 6), 8),
    ("greedycd", np.float64, 2, (300, 530, 6), 6),
    ("cd", np.float64, 8, (300, 530, 6), 6),
])
def test_peer_windows_across_processes_on_one_device(built, alg, T, G, shape, iters):
    """One PROCESS per rank, all on device 0, windows mapped through hipIpc: bit-identical to the in-process group."""
    p, n, k = shape
    X, W0, H0 = planted(p, n, k, T, seed=17, normalize=(alg != "projals"))
    lam = lam_for(alg, T)
    kw = dict(maxiter=iters, tol=1e-30, lambda_w=lam, lambda_h=lam, track_objective=True)
    Wr, Hr, rr, Wall = run_sharded(T, X, W0, H0, alg, kw, G)
    res = run_peer_processes(T, shape, alg, G, iters, lam)
    for r, d in enumerate(res):
        assert np.array_equal(d["W"], Wr), f"rank {r}: W differs from the in-process group's"
        assert np.array_equal(d["H"], Hr[:, int(d["c0"]):int(d["c1"])])
        assert np.array_equal(d["trace"], np.asarray(rr[r][1]), equal_nan=True)
        assert int(d["niters"]) == rr[r][0].niters
        if alg == "alspgrad":
            assert int(d["inner"]) == rr[r][0].inner_iters and int(d["backtracks"]) == rr[r][0].backtracks
        assert int(d["served"]) > 0 and int(d["based"]) == 0


def test_peer_timeout_is_an_error_not_a_hang(built):
    """A rank that never arrives: the survivor's wait gives up after NMFX_P2P_TIMEOUT_S and the solve fails with a communicator
    error (no spinning kernel is left on the GPU)."""
    T = np.float32
    p, n, k = 256, 300, 4
    X, W0, H0 = planted(p, n, k, T, seed=5)
    os.environ["NMFX_P2P_TIMEOUT_S"] = "1.5"
    try:
        with nmfx.Context(T, p, 150, k) as a, nmfx.Context(T, p, 150, k) as b:
            a.comm_init_p2p(0, 2)
            b.comm_init_p2p(1, 2)
            hs = [a.comm_p2p_export(), b.comm_p2p_export()]
            a.comm_p2p_attach(hs)
            b.comm_p2p_attach(hs)
            a.set_X(np.asfortranarray(X[:, :150]))
            W, H = W0.copy(order="F"), np.asfortranarray(H0[:, :150].copy())
            with pytest.raises(Exception, match="timed out"):
                a.solve(ALG["multmse"], nmfx.make_opts(T, maxiter=2, tol=1e-30), W, H)      # rank 1 never runs
    finally:
        del os.environ["NMFX_P2P_TIMEOUT_S"]


def test_peer_timeout_fails_every_rank_including_one_that_was_only_slow(built):
    """a rank whose wait times out raises the abort word in EVERY mapped window.  Rank 1 here is merely late (it starts
    its solve well after rank 0 gave up): without the propagation it would find rank 0's later flags, sum slot data of later
    collectives and return NMFX_OK with corrupted factors.  Both solves must end with the communicator error."""
    import time
    T = np.float32
    p, n, k = 256, 300, 4
    X, W0, H0 = planted(p, n, k, T, seed=5)
    os.environ["NMFX_P2P_TIMEOUT_S"] = "1.0"
    errs = [None, None]
    try:
        with nmfx.Context(T, p, 150, k) as a, nmfx.Context(T, p, 150, k) as b:
            a.comm_init_p2p(0, 2)
            b.comm_init_p2p(1, 2)
            hs = [a.comm_p2p_export(), b.comm_p2p_export()]
            a.comm_p2p_attach(hs)
            b.comm_p2p_attach(hs)
            a.set_X(np.asfortranarray(X[:, :150]))
            b.set_X(np.asfortranarray(X[:, 150:]))

            def run(ctx, r, delay, c0, c1):
                time.sleep(delay)
                W, H = W0.copy(order="F"), np.asfortranarray(H0[:, c0:c1].copy())
                try:
                    ctx.solve(ALG["multmse"], nmfx.make_opts(T, maxiter=6, tol=1e-30), W, H)
                    errs[r] = "returned without an error"
                except Exception as e:  # noqa: BLE001
                    errs[r] = repr(e)

            th = [threading.Thread(target=run, args=(a, 0, 0.0, 0, 150)), threading.Thread(target=run, args=(b, 1, 4.0, 150, 300))]
            for t in th:
                t.start()
            for t in th:
                t.join(120)
    finally:
        del os.environ["NMFX_P2P_TIMEOUT_S"]
    assert errs[0] is not None and "timed out" in errs[0], errs
    assert errs[1] is not None and "timed out" in errs[1], errs
