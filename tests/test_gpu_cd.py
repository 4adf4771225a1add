"""GPU parity: CoordinateDescent (src/coorddesc.jl) and GreedyCD (src/greedycd.jl) through the C ABI vs the CPU oracles
(SURVEY.md section 8f rank 2).

Tolerances on the objective trajectory: CD 1e-9 (f64) / 5e-4 (f32) relative; GreedyCD 1e-9 (f64) / 3e-3 (f32).  CD: the device forms the row dot products as
a butterfly instead of left to right.  GreedyCD: the per-row greedy sweep is restated operation by operation (no FMA
contraction), but its inputs G = W*P - Z come from MFMA GEMMs whose rounding differs from a CPU GEMM, and the sweep is
discontinuous in them (argmax / threshold decisions) -- like alspgrad, trajectories are compared, not decision traces; in
f64 the executed greedy step count equals the oracle's on these inputs.
"""
import numpy as np
import pytest

import c_oracle as co
import nmf_oracle as orc
import nmfx
from problems import planted, rel_trace_err, uniform

pytestmark = pytest.mark.gpu
TOL = {np.float64: 1e-9, np.float32: 5e-4}
TOL_GREEDY = {np.float64: 1e-9, np.float32: 3e-3}    # f32: the sweep's argmax / threshold decisions flip on 1-ulp input changes
SHAPES = [(64, 96, 5), (300, 260, 70), (130, 515, 8), (129, 257, 100), (7, 5, 5), (33, 1000, 3), (257, 300, 130)]


def _cd(T, **kw):
    return nmfx.CoordinateDescent(T, **kw)


@pytest.mark.parametrize("T", [np.float64, np.float32])
@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("reg", [False, True])
def test_cd_trajectory(built, T, shape, reg):
    p, n, k = shape
    X, W0, H0 = uniform(p, n, k, T, seed=5 + p)
    kw = dict(alpha=2e-3, l1ratio=0.25, regularization="both") if reg else {}
    alg = _cd(T, maxiter=8, tol=1e-30, **kw)
    Wg, Hg = W0.copy(order="F"), H0.copy(order="F")
    r = nmfx.solve(alg, X, Wg, Hg, track_objective=True)
    oo = orc.Opts(maxiter=8, tol=1e-30, track_objective=True, l1_w=alg.l1_w, l2_w=alg.l2_w, l1_h=alg.l1_h, l2_h=alg.l2_h)
    Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
    ro = (orc if p * n * k < 2e6 else co).solve("cd", X, Wc, Hc, oo)
    assert r.niters == ro.niters == 8
    assert rel_trace_err(r.trace, ro.trace) < TOL[T]
    assert np.all(Wg >= 0) and np.all(Hg >= 0)
    assert np.max(np.abs(Wg - Wc)) <= 200 * TOL[T] * max(1.0, np.max(np.abs(Wc)))
    assert np.max(np.abs(Hg - Hc)) <= 200 * TOL[T] * max(1.0, np.max(np.abs(Hc)))


@pytest.mark.parametrize("T", [np.float64, np.float32])
@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("lam", [(0.0, 0.0), (1e-3, 2e-3)])
def test_greedycd_trajectory(built, T, shape, lam):
    p, n, k = shape
    X, W0, H0 = uniform(p, n, k, T, seed=11 + n)
    alg = nmfx.GreedyCD(T, maxiter=6, tol=1e-30, lambda_w=lam[0], lambda_h=lam[1])
    Wg, Hg = W0.copy(order="F"), H0.copy(order="F")
    r = nmfx.solve(alg, X, Wg, Hg, track_objective=True)
    oo = orc.Opts(maxiter=6, tol=1e-30, track_objective=True, lambda_w=lam[0], lambda_h=lam[1])
    Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
    ro = co.solve("greedycd", X, Wc, Hc, oo)
    assert r.niters == ro.niters == 6
    # conditioning yardstick (as for projals): how far the SAME CPU restatement drifts when X moves by one ulp in half of
    # its entries -- with k close to min(p, n) the objective is not even monotone and the f32 trajectories are chaotic
    # at the 1e-2 level.  The device may deviate from the oracle by at most 3x that, nor is it asked to beat it.
    rng = np.random.default_rng(1)
    Xp = np.asfortranarray(np.where(rng.random(X.shape) < 0.5, np.nextafter(X, T(2)), X).astype(T))
    ry = co.solve("greedycd", Xp, W0.copy(order="F"), H0.copy(order="F"), oo)
    tol = max(TOL_GREEDY[T], 3 * rel_trace_err(ry.trace, ro.trace))
    err = rel_trace_err(r.trace, ro.trace)
    if err >= tol and T == np.float32:
        # second yardstick, only when needed (slow): the two independent CPU restatements (NumPy/OpenBLAS GEMMs vs the plain-C
        # loops) on the SAME input -- e.g. 1.3e-2 apart at (129, 257, 100), where the device is 1.2e-2 from the C one
        rn = orc.solve("greedycd", X, W0.copy(order="F"), H0.copy(order="F"), oo)
        tol = max(tol, 3 * rel_trace_err(rn.trace, ro.trace))
    assert err < tol
    assert np.all(Wg >= 0) and np.all(Hg >= 0)
    if T == np.float64:
        assert r.info["inner_iters"] == ro.counters["inner"]           # executed greedy steps
        assert np.max(np.abs(Wg - Wc)) <= 1e-7 * max(1.0, np.max(np.abs(Wc)))
        assert np.max(np.abs(Hg - Hc)) <= 1e-7 * max(1.0, np.max(np.abs(Hc)))


@pytest.mark.parametrize("T", [np.float64, np.float32])
def test_reference_kats(built, T):
    """test/coorddesc.jl:5-8 and test/greedycd.jl:5-20 on the device path (laurberg6x3, W perturbed, H = Hg)."""
    rng = np.random.default_rng(3)
    X, Wg, Hg = orc.laurberg6x3(0.3, T)
    W = np.asfortranarray(Wg + rng.random(Wg.shape).astype(T) * T(0.1)); H = Hg.copy(order="F")
    nmfx.solve(_cd(T, alpha=0.0, maxiter=1000, tol=1e-9), X, W, H)
    assert np.allclose(X, W @ H, atol=1e-4, rtol=0)
    W = np.asfortranarray(Wg + rng.random(Wg.shape).astype(T) * T(0.1)); H = Hg.copy(order="F")
    nmfx.solve(_cd(T, alpha=1e-4, l1ratio=0.5, maxiter=1000, tol=1e-9), X, W, H)      # (the reference also shuffles here)
    assert np.allclose(X, W @ H, atol=1e-2, rtol=0)
    for lw in (0.0, 1e-5):
        for lh in (0.0, 1e-5):
            W = np.asfortranarray(Wg + rng.random(Wg.shape).astype(T) * T(0.1)); H = Hg.copy(order="F")
            nmfx.solve(nmfx.GreedyCD(T, maxiter=1000, tol=1e-9, lambda_w=lw, lambda_h=lh), X, W, H)
            assert np.all(W >= 0) and np.all(H >= 0) and not np.isnan(W).any() and not np.isnan(H).any()
            assert np.allclose(X, W @ H, atol=1e-3, rtol=0)


@pytest.mark.parametrize("algname", ["cd", "greedycd"])
def test_update_H_false_and_stop(built, algname):
    """test/interf.jl:33-37 for :cd / :greedycd: update_H=false leaves H bit-identical; the stop rule matches the oracle."""
    T = np.float64
    X, W0, H0 = planted(60, 80, 4, T, seed=8)
    mk = (lambda **kw: _cd(T, **kw)) if algname == "cd" else (lambda **kw: nmfx.GreedyCD(T, **kw))
    Wg, Hg = W0.copy(order="F"), H0.copy(order="F")
    nmfx.solve(mk(maxiter=20, update_H=False), X, Wg, Hg)
    assert np.array_equal(Hg, H0) and not np.array_equal(Wg, W0)
    Wg, Hg = W0.copy(order="F"), H0.copy(order="F")
    r = nmfx.solve(mk(maxiter=500, tol=1e-5), X, Wg, Hg)
    ro = orc.solve(algname, X, W0.copy(order="F"), H0.copy(order="F"), orc.Opts(maxiter=500, tol=1e-5))
    assert r.converged == ro.converged and abs(r.niters - ro.niters) <= 1
    assert abs(r.objvalue - ro.objvalue) <= 1e-7 * abs(ro.objvalue)


@pytest.mark.parametrize("algname", ["cd", "greedycd"])
def test_sharded_path_single_rank(built, algname):
    """The RCCL code path (packed all-reduce, p_init max all-reduce, H statistics all-reduce) with a 1-rank communicator."""
    T = np.float32
    X, W0, H0 = uniform(200, 150, 6, T, seed=2)
    alg = _cd(T, maxiter=5, tol=1e-30) if algname == "cd" else nmfx.GreedyCD(T, maxiter=5, tol=1e-30)
    W1, H1 = W0.copy(order="F"), H0.copy(order="F")
    r1 = nmfx.solve(alg, X, W1, H1)
    with nmfx.Context(T, 200, 150, 6) as ctx:
        ctx.set_X(X)
        ctx.comm_init(nmfx.comm_unique_id(), 0, 1)
        W2, H2 = W0.copy(order="F"), H0.copy(order="F")
        r2 = nmfx.solve(alg, X, W2, H2, ctx=ctx)
    assert r1.objvalue == r2.objvalue and np.array_equal(W1, W2) and np.array_equal(H1, H2)


def test_nnmf_default_algorithm(built):
    """nnmf's default alg is :greedycd (src/interf.jl:6)."""
    X, _, _ = uniform(50, 70, 4, np.float64, seed=4)
    r = nmfx.nnmf(X, 4, init="random", alg="greedycd", maxiter=30, rng=np.random.default_rng(0))
    assert np.isfinite(r.objvalue) and (r.W >= 0).all() and (r.H >= 0).all()
    r2 = nmfx.nnmf(X, 4, init="random", alg="cd", maxiter=30, seed=3, replicates=2)
    assert np.isfinite(r2.objvalue)


def test_k_limit(built):
    """k > 1024 runs the LDS forms of the sweeps (test_sweeps_beyond_1024_components); what is refused is a sample row whose
    component vectors exceed 160 KiB of LDS (GreedyCD: k > ~4400 in Float32)."""
    k = 4608
    X, W0, H0 = uniform(k + 8, k + 8, k, np.float32, seed=1)
    with pytest.raises(nmfx.NMFXError, match="k too large"):
        nmfx.solve(nmfx.GreedyCD(np.float32, maxiter=2), X, W0, H0)


@pytest.mark.parametrize("T", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(64, 96, 5), (130, 515, 8), (300, 260, 70), (129, 257, 100), (200, 180, 130)])
@pytest.mark.parametrize("update_H", [True, False])
def test_cd_shuffle_matches_oracle(built, T, shape, update_H):
    """CoordinateDescent(shuffle = true) (src/coorddesc.jl:130-131): a fresh component order per side and iteration.  Device and
    oracle use the same documented orders (include/nmfx.h; tests/philox_ref.py::cd_permutation), so the trajectories must agree
    like they do for shuffle = false -- and differ from the shuffle = false run."""
    import philox_ref
    p, n, k = shape
    X, W0, H0 = planted(p, n, k, T, seed=40 + k)
    seed = -123456 if k == 8 else 77 + k                          # (a negative key exercises the sign extension)
    inst = nmfx.CoordinateDescent(T, maxiter=6, tol=1e-30, alpha=1e-3, l1ratio=0.5, shuffle=True, shuffle_seed=seed, update_H=update_H)
    W, H = W0.copy(order="F"), H0.copy(order="F")
    r = nmfx.solve(inst, X, W, H, track_objective=True)
    Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
    o = orc.Opts(maxiter=6, tol=1e-30, update_H=update_H, l1_w=inst.l1_w, l2_w=inst.l2_w, l1_h=inst.l1_h, l2_h=inst.l2_h, track_objective=True,
                 perm_source=lambda c: philox_ref.cd_permutation(k, seed, c))
    ro = orc.solve("cd", X, Wc, Hc, o)
    assert r.niters == ro.niters == 6
    tol = {np.float64: 1e-9, np.float32: 5e-4}[T]
    assert rel_trace_err(r.trace, ro.trace) < tol
    assert np.max(np.abs(W - Wc)) <= 200 * tol * np.max(np.abs(Wc))
    if update_H:
        assert np.max(np.abs(H - Hc)) <= 200 * tol * np.max(np.abs(Hc))
    else:
        assert np.array_equal(H, H0)
    W2, H2 = W0.copy(order="F"), H0.copy(order="F")
    nmfx.solve(nmfx.CoordinateDescent(T, maxiter=6, tol=1e-30, alpha=1e-3, l1ratio=0.5, update_H=update_H), X, W2, H2)
    assert not np.array_equal(W, W2)


def _solve_with_env(env, inst, X, W0, H0):
    import os
    old = os.environ.get("NMFX_CD_LDS")
    if env:
        os.environ["NMFX_CD_LDS"] = "1"
    try:
        W, H = W0.copy(order="F"), H0.copy(order="F")
        with nmfx.Context(inst.T, X.shape[0], X.shape[1], W0.shape[1]) as ctx:      # the knob is read when the context is created
            ctx.set_X(X)
            r = nmfx.solve(inst, X, W, H, ctx=ctx, track_objective=True)
    finally:
        if old is None:
            os.environ.pop("NMFX_CD_LDS", None)
        else:
            os.environ["NMFX_CD_LDS"] = old
    return W, H, r


@pytest.mark.parametrize("T", [np.float64, np.float32])
@pytest.mark.parametrize("alg", ["cd", "greedycd"])
@pytest.mark.parametrize("shape", [(90, 120, 70), (140, 100, 130), (64, 64, 64), (640, 660, 600), (300, 280, 256), (230, 210, 192), (260, 270, 250)])   # every slot count of the all-slots-live greedy form (1..4), full and ragged last slots
def test_lds_forms_of_the_sweeps_are_bit_identical(built, T, alg, shape):
    """k > 1024 runs the sweeps with the sample row's component vectors in LDS (one wave per row) instead of registers.  Same
    expressions, same accumulation order as the 64-lanes-per-row register kernels: forced at small k (NMFX_CD_LDS=1) the two forms
    must agree bit for bit -- factors, objective trajectory and (greedycd) the executed step count.  (CoordinateDescent at small k
    normally runs its 16-lanes-per-row kernel, whose dot products are summed in another order: there the comparison is to rounding.)"""
    p, n, k = shape
    X, W0, H0 = uniform(p, n, k, T, seed=3 + k)
    iters = 5 if k < 500 else 2
    inst = _cd(T, maxiter=iters, tol=1e-30, alpha=1e-3, l1ratio=0.5, regularization="both") if alg == "cd" else \
        nmfx.GreedyCD(T, maxiter=iters, tol=1e-30, lambda_w=1e-3, lambda_h=2e-3)
    Wr, Hr, rr = _solve_with_env(False, inst, X, W0, H0)
    Wl, Hl, rl = _solve_with_env(True, inst, X, W0, H0)
    K = 64 if k <= 64 else (k + 127) // 128 * 128
    same_kernel_order = alg == "greedycd" or K // 16 > (32 if T == np.float32 else 16)     # cd: beyond cd_sweep16_kernel's register budget
    if same_kernel_order:
        assert np.array_equal(Wr, Wl) and np.array_equal(Hr, Hl) and np.array_equal(rr.trace, rl.trace)
    else:
        tol = 1e-11 if T == np.float64 else 1e-4
        assert rel_trace_err(rl.trace, rr.trace) < tol
        assert np.max(np.abs(Wr - Wl)) <= 100 * tol * max(1.0, np.max(np.abs(Wr))) and np.max(np.abs(Hr - Hl)) <= 100 * tol * max(1.0, np.max(np.abs(Hr)))
    if alg == "greedycd":
        assert rr.info["inner_iters"] == rl.info["inner_iters"] > 0


@pytest.mark.parametrize("shape", [(90, 120, 5), (200, 130, 64), (300, 260, 70), (240, 200, 130), (400, 450, 300), (600, 700, 500)])   # k < min(p, n): a rank-deficient Gram makes the f32 sweep chaotic on the CPU too
@pytest.mark.parametrize("variant", ["plain", "l1", "w_only", "shuffle"])
@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_cd_blocked_sweep_matches_the_row_chain_sweep(built, T, shape, variant, monkeypatch):
    """CoordinateDescent with k <= 512 sweeps 16 coordinates at a time (Float64 beyond k = 256: 32-row tiles): the gradient of the block from one matrix-core product
    per 64 sample rows, the sweep inside the block on the 16 x 16 diagonal block of the Gram (cd.hpp: cd_sweep_blocked_kernel) -- the
    same Gauss-Seidel sweep in exact arithmetic, another summation order.  Against the row-chain kernels (NMFX_CD_BLOCKED=0) and
    against the CPU oracle: objective trajectories to the f32 tolerance of this file, factors to rounding; padded components
    (k not a multiple of 16 / of the padded width 64, 128, 256, 384, 512) stay inert."""
    p, n, k = shape
    X, W0, H0 = uniform(p, n, k, T, seed=11 + k)
    kw = dict(maxiter=4, tol=1e-30)
    if variant == "l1":
        kw.update(alpha=1e-3, l1ratio=0.5, regularization="both")
    if variant == "w_only":
        kw.update(update_H=False)
    if variant == "shuffle":
        kw.update(shuffle=True, shuffle_seed=5)
    inst = _cd(T, **kw)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("NMFX_CD_BLOCKED", mode)
        W, H = W0.copy(order="F"), H0.copy(order="F")
        with nmfx.Context(T, p, n, k) as ctx:      # the knob is read when the context is created
            ctx.set_X(X)
            out[mode] = (nmfx.solve(inst, X, W, H, ctx=ctx, track_objective=True), W, H)
    (rb, Wb, Hb), (rc, Wc, Hc) = out["1"], out["0"]
    assert rb.niters == rc.niters == 4
    assert rel_trace_err(rb.trace, rc.trace) < TOL[T]
    ftol = 2e-3 if T == np.float32 else 1e-8
    assert np.max(np.abs(Wb - Wc)) <= ftol * max(1.0, np.max(np.abs(Wc))) and np.max(np.abs(Hb - Hc)) <= ftol * max(1.0, np.max(np.abs(Hc)))
    assert np.all(Wb >= 0) and np.all(Hb >= 0)
    if variant == "w_only":
        assert np.array_equal(Hb, H0)
    if variant != "shuffle":       # (the shuffled order has its own oracle test above)
        Wo, Ho = W0.copy(order="F"), H0.copy(order="F")
        oo = orc.Opts(maxiter=4, tol=1e-30, track_objective=True, update_H=(variant != "w_only"),
                      l1_w=inst.l1_w, l2_w=inst.l2_w, l1_h=inst.l1_h, l2_h=inst.l2_h)
        ro = co.solve("cd", X, Wo, Ho, oo)
        assert rel_trace_err(rb.trace, ro.trace) < TOL[T]


@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_sweeps_beyond_1024_components(built, T):
    """The reference's sweeps have no size limit (src/coorddesc.jl:133-156, src/greedycd.jl:134-158); round 2 refused k > 1024.
    CoordinateDescent against the C oracle at k = 1100, GreedyCD through its properties (exact coordinate minimisation never
    increases the objective; non-negative, finite factors; the greedy steps are counted)."""
    p, n, k = 1150, 1210, 1100
    X, W0, H0 = planted(p, n, k, T, seed=8, k0=20)
    alg = _cd(T, maxiter=2, tol=1e-30)
    Wg, Hg = W0.copy(order="F"), H0.copy(order="F")
    r = nmfx.solve(alg, X, Wg, Hg, track_objective=True)
    Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
    ro = co.solve("cd", X, Wc, Hc, orc.Opts(maxiter=2, tol=1e-30, track_objective=True))
    tol = 1e-9 if T == np.float64 else 2e-3
    assert r.niters == ro.niters == 2 and rel_trace_err(r.trace, ro.trace) < tol
    assert np.max(np.abs(Wg - Wc)) <= 200 * tol * max(1.0, np.max(np.abs(Wc)))
    W2, H2 = W0.copy(order="F"), H0.copy(order="F")
    r2 = nmfx.solve(nmfx.GreedyCD(T, maxiter=2, tol=1e-30), X, W2, H2, track_objective=True)
    assert r2.niters == 2 and np.all(np.diff(r2.trace) <= 1e-6 * r2.trace[0])
    assert np.all(W2 >= 0) and np.all(H2 >= 0) and np.isfinite(W2).all() and np.isfinite(H2).all()
    assert r2.info["inner_iters"] > 0


def test_cd_shuffle_orders_beyond_one_window(built):
    """The component orders are generated a window of 256 iterations at a time (all 2 * maxiter * k orders used
    to be built before the first iteration): a solve that crosses a window boundary still follows the oracle's orders, and a huge
    maxiter with a tolerance that stops the solve early costs one window."""
    import philox_ref
    T = np.float64
    p, n, k = 40, 56, 4
    X, W0, H0 = planted(p, n, k, T, seed=12)
    seed, iters = 91, 300
    inst = nmfx.CoordinateDescent(T, maxiter=iters, tol=1e-30, alpha=1e-3, l1ratio=0.5, shuffle=True, shuffle_seed=seed)
    W, H = W0.copy(order="F"), H0.copy(order="F")
    r = nmfx.solve(inst, X, W, H)
    Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
    o = orc.Opts(maxiter=iters, tol=1e-30, l1_w=inst.l1_w, l2_w=inst.l2_w, l1_h=inst.l1_h, l2_h=inst.l2_h,
                 perm_source=lambda c: philox_ref.cd_permutation(k, seed, c))
    ro = orc.solve("cd", X, Wc, Hc, o)
    assert r.niters == ro.niters == iters
    assert abs(r.objvalue - ro.objvalue) <= 1e-9 * abs(ro.objvalue)
    assert np.max(np.abs(W - Wc)) <= 1e-7 * np.max(np.abs(Wc)) and np.max(np.abs(H - Hc)) <= 1e-7 * np.max(np.abs(Hc))
    W2, H2 = W0.copy(order="F"), H0.copy(order="F")
    r2 = nmfx.solve(nmfx.CoordinateDescent(T, maxiter=2_000_000_000, tol=1e-3, alpha=1e-3, l1ratio=0.5, shuffle=True, shuffle_seed=seed), X, W2, H2)
    assert r2.converged and r2.niters < 2000


def test_cd_shuffle_reference_kat(built):
    """test/coorddesc.jl:10-14 on the GPU path: alpha = 1e-4, l1ratio = 0.5, shuffle = true reconstructs X to 1e-2."""
    for T in (np.float64, np.float32):
        X, Wg, Hg = orc.laurberg6x3(0.3, T)
        W = np.asfortranarray(Wg + np.random.default_rng(2).random(Wg.shape).astype(T) * T(0.1))
        H = Hg.copy(order="F")
        nmfx.solve(nmfx.CoordinateDescent(T, alpha=1e-4, l1ratio=0.5, shuffle=True, maxiter=1000, tol=1e-9), X, W, H)
        assert np.allclose(X, W @ H, atol=1e-2, rtol=0)



@pytest.mark.parametrize("shape", [(40, 50, 5), (90, 70, 64), (100, 120, 70), (130, 140, 150), (150, 160, 192), (210, 220, 250), (200, 210, 256), (140, 150, 300)])
@pytest.mark.parametrize("lam", [0.0, 0.25])
@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_greedy_sweep_arithmetic_is_bit_identical_to_the_oracle(built, T, shape, lam):
    """The Float32 trajectories above are compared to a tolerance because the sweep's INPUTS differ in the last bit: G = W*P - Z
    comes out of MFMA GEMMs that sum in another order than a CPU GEMM, and the sweep is discontinuous in them.  With small-integer
    X, W0, H0 every product and sum of P = H*H', Z = X*H', G = W*P - Z is exact in Float32 whatever the order, so both sides
    sweep the SAME numbers -- and then the W side of the first iteration (update_H = false: nothing else runs) must agree BIT FOR
    BIT with the C restatement of src/greedycd.jl:91-163: the same divisions, the same separately rounded products and sums, the
    same first-index arg-max, the same number of greedy steps.  Shapes: every slot count of the register forms (k <= 64, 128,
    192, 256, full and ragged last slots) and the general form beyond."""
    p, n, k = shape
    rng = np.random.default_rng(100 + k)
    X = np.asfortranarray(rng.integers(0, 4, size=(p, n)).astype(T))
    W0 = np.asfortranarray(rng.integers(0, 3, size=(p, k)).astype(T))
    H0 = np.asfortranarray(rng.integers(0, 3, size=(k, n)).astype(T))
    assert 2 * 2 * k * 2 * 2 * n < 2 ** 24        # |W*P| <= k * 2 * (n * 4): exact in Float32
    o = nmfx.make_opts(T, maxiter=1, tol=1e-30, update_H=False, lambda_w=lam, lambda_h=lam, check_every=1000)
    Wg, Hg = W0.copy(order="F"), H0.copy(order="F")
    with nmfx.Context(T, p, n, k) as ctx:
        ctx.set_X(X)
        ctx.set_factors(Wg, Hg)
        res, _ = ctx.iterate(5, o)
        ctx.get_factors(Wg, Hg)
    Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
    ro = co.solve("greedycd", X, Wc, Hc, orc.Opts(maxiter=1, tol=1e-30, update_H=False, lambda_w=lam, lambda_h=lam))
    assert res.niters == ro.niters == 1
    assert np.array_equal(Hg, H0) and np.array_equal(Hc, H0)
    assert res.inner_iters == ro.counters["inner"] > 0
    U = np.uint32 if T == np.float32 else np.uint64
    assert np.array_equal(Wg.view(U), Wc.view(U)), float(np.max(np.abs(Wg - Wc)))
    assert not np.array_equal(Wg, W0)


@pytest.mark.parametrize("shape", [(40, 50, 5), (300, 90, 64), (500, 170, 100), (700, 300, 256), (400, 700, 600)])
@pytest.mark.parametrize("reg", [(0.0, 0.0), (0.25, 0.5)])
@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_cd_sweep_arithmetic_is_bit_identical_when_the_gram_is_diagonal(built, T, shape, reg):
    """CoordinateDescent's W side (src/coorddesc.jl:107-158) with an H0 whose rows are indicators of distinct columns: P = HH' = I
    and Z = XH' is a column selection of X, both exact, and every dot product of the sweep has a single non-zero term -- the
    summation order (butterfly / MFMA block products on the device, left to right in the reference) cannot matter, so the
    element-wise rule w <- max(0, w - (g + l1) / (P_tt + l2)) must give the oracle's W BIT FOR BIT (update_H = false: nothing
    else runs).  Covers the 16-lane, 64-lane, blocked and LDS-resident forms by k."""
    p, n, k = shape
    rng = np.random.default_rng(17 + k)
    X = np.asfortranarray(rng.integers(0, 9, size=(p, n)).astype(T))
    W0 = np.asfortranarray((rng.integers(0, 40, size=(p, k)) / 8.0).astype(T))      # dyadic start values
    cols = rng.permutation(n)[:k]
    H0 = np.zeros((k, n), dtype=T, order="F")
    H0[np.arange(k), cols] = 1
    l1, l2 = reg
    o = nmfx.make_opts(T, maxiter=1, tol=1e-30, update_H=False, l1_w=l1, l2_w=l2, l1_h=l1, l2_h=l2, check_every=1000)
    Wg, Hg = W0.copy(order="F"), H0.copy(order="F")
    with nmfx.Context(T, p, n, k) as ctx:
        ctx.set_X(X)
        ctx.set_factors(Wg, Hg)
        ctx.iterate(4, o)
        ctx.get_factors(Wg, Hg)
    Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
    co.solve("cd", X, Wc, Hc, orc.Opts(maxiter=1, tol=1e-30, update_H=False, l1_w=l1, l2_w=l2, l1_h=l1, l2_h=l2))
    U = np.uint32 if T == np.float32 else np.uint64
    assert np.array_equal(Wg.view(U), Wc.view(U)), float(np.max(np.abs(Wg - Wc)))
    assert not np.array_equal(Wg, W0) and np.array_equal(Hg, H0)
