"""GPU parity: MultUpdate (MSE and divergence) through the C ABI vs the CPU oracle.

Tolerances (stated per north_star): objective trajectory within 1e-5 relative (f32) /
1e-10 (f64) iteration for iteration; final W, H within 1e-3 / 1e-8 relative to max|.|.
The reference forms W'(WH) and (WH)H' through the p x n product, the GPU path uses the
Gram form (W'W)H and W(HH'): algebraically identical, differs in rounding only.
"""
import numpy as np
import pytest

import nmf_oracle as orc
import nmfx
from problems import planted, rel_trace_err, uniform

pytestmark = pytest.mark.gpu

# includes degenerate and tile-boundary shapes: k = 1, p <= 256 with k = 128 (one k-tile column: the fused-Gram tail path
# with a single main split), k just above a tile multiple
SHAPES = [(2, 3, 1), (6, 6, 3), (5, 8, 3), (200, 500, 5), (300, 260, 70), (512, 768, 64), (257, 1030, 129),
          (129, 257, 128)]
TOL = {np.float32: (1e-5, 1e-3), np.float64: (1e-10, 1e-8)}


@pytest.mark.parametrize("T", [np.float32, np.float64])
@pytest.mark.parametrize("obj", ["mse", "div"])
@pytest.mark.parametrize("shape", SHAPES)
def test_trajectory_matches_oracle(built, T, obj, shape):
    p, n, k = shape
    X, W0, H0 = planted(p, n, k, T, seed=p * 1000 + n)
    maxiter = 25
    alg = nmfx.MultUpdate(T, obj=obj, maxiter=maxiter, tol=1e-30, lambda_w=1e-4, lambda_h=1e-4)
    Wg, Hg = W0.copy(order="F"), H0.copy(order="F")
    r = nmfx.solve(alg, X, Wg, Hg, track_objective=True)
    Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
    ro = orc.solve("mult" + obj, X, Wc, Hc, orc.Opts(maxiter=maxiter, tol=1e-30, lambda_w=1e-4, lambda_h=1e-4,
                                                     track_objective=True))
    assert r.niters == ro.niters == maxiter and not r.converged
    tol_obj, tol_fac = TOL[T]
    # relative to each point, plus the cancellation floor of the objective itself: degenerate fits (1x1, 2x3, k=1)
    # drive the objective from O(1) to ~1e-6 of its start, where the sum of O(1) terms only carries ~eps(T) of
    # absolute accuracy -- the two CPU restatements differ there by the same amount.
    tr, tro = np.asarray(r.trace), np.asarray(ro.trace)
    assert np.all(np.abs(tr - tro) <= tol_obj * np.abs(tro) + 8 * np.finfo(T).eps * abs(tro[0]))
    assert np.max(np.abs(Wg - Wc)) <= tol_fac * np.max(np.abs(Wc))
    assert np.max(np.abs(Hg - Hc)) <= tol_fac * np.max(np.abs(Hc))
    assert np.all(Wg >= 0) and np.all(Hg >= 0) and not np.isnan(Wg).any() and not np.isnan(Hg).any()


@pytest.mark.parametrize("T", [np.float32, np.float64])
@pytest.mark.parametrize("obj", ["mse", "div"])
def test_stop_rule_same_iteration(built, T, obj):
    """converged / niters agree with the oracle when the stop rule fires (src/common.jl:92-111)."""
    X, W0, H0 = planted(120, 90, 4, T, seed=5)
    tol = 2e-2
    alg = nmfx.MultUpdate(T, obj=obj, maxiter=400, tol=tol)
    Wg, Hg = W0.copy(order="F"), H0.copy(order="F")
    r = nmfx.solve(alg, X, Wg, Hg, check_every=3)
    Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
    ro = orc.solve("mult" + obj, X, Wc, Hc, orc.Opts(maxiter=400, tol=tol))
    assert ro.converged and r.converged
    assert r.niters == ro.niters
    assert abs(r.objvalue - ro.objvalue) <= TOL[T][0] * 10 * abs(ro.objvalue)
    assert np.max(np.abs(Wg - Wc)) <= TOL[T][1] * np.max(np.abs(Wc))


@pytest.mark.parametrize("T", [np.float32, np.float64])
@pytest.mark.parametrize("obj", ["mse", "div"])
def test_update_H_false_leaves_H_bit_identical(built, T, obj):
    """test/interf.jl:33-37."""
    X, W0, H0 = uniform(40, 64, 3, T, seed=3)
    alg = nmfx.MultUpdate(T, obj=obj, maxiter=30, update_H=False)
    W, H = W0.copy(order="F"), H0.copy(order="F")
    r = nmfx.solve(alg, X, W, H)
    assert np.array_equal(H, H0) and r.H is H
    assert np.any(W != W0)
    Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
    ro = orc.solve("mult" + obj, X, Wc, Hc, orc.Opts(maxiter=30, update_H=False,
                                                     tol=float(T(np.cbrt(np.finfo(T).eps)))))
    assert r.niters == ro.niters
    assert np.max(np.abs(W - Wc)) <= TOL[T][1] * np.max(np.abs(Wc))


@pytest.mark.parametrize("T", [np.float64, np.float32])
@pytest.mark.parametrize("obj", ["mse", "div"])
@pytest.mark.parametrize("lam", [0.0, 1e-4])
def test_reference_kat_laurberg(built, T, obj, lam):
    """test/multupd.jl:3-22 run on the GPU path: non-negative, no NaN, ||X - W*Hg||_F <= 1e-2."""
    X, Wg, Hg = orc.laurberg6x3(0.3, T)
    rng = np.random.default_rng(11)
    W = np.asfortranarray(Wg + rng.random(Wg.shape).astype(T) * T(0.1))
    H = Hg.copy(order="F")
    alg = nmfx.MultUpdate(T, obj=obj, maxiter=5000, tol=1e-9, lambda_w=lam, lambda_h=lam)
    nmfx.solve(alg, X, W, H, check_every=64)
    assert np.all(W >= 0) and np.all(H >= 0)
    assert not np.isnan(W).any() and not np.isnan(H).any()
    assert np.linalg.norm(X - W @ H) <= 1e-2


def test_argument_errors(built):
    """Constructor validation of src/multupd.jl:27-31 and the shape check of src/common.jl:5-16."""
    T = np.float32
    with pytest.raises(nmfx.ArgumentError):
        nmfx.MultUpdate(T, obj="foo")
    with pytest.raises(nmfx.ArgumentError):
        nmfx.MultUpdate(T, maxiter=1)
    with pytest.raises(nmfx.ArgumentError):
        nmfx.MultUpdate(T, tol=0.0)
    with pytest.raises(nmfx.ArgumentError):
        nmfx.MultUpdate(T, lambda_w=-1.0)
    X, W0, H0 = uniform(12, 10, 3, T)
    with pytest.raises(nmfx.DimensionMismatch):
        nmfx.solve(nmfx.MultUpdate(T), X, W0, np.asfortranarray(H0[:, :5]))


@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_multdiv_raw_abi_lambda_zero_padded_k(built, T):
    """A raw C-ABI caller (make_opts defaults lambda_w = lambda_h = 0) on a k that is padded on the device (5 -> 64):
    the padded components must stay inert (0 * (0/0) would poison W, H and the objective) and the library applies the
    sqrt(eps(T)) floor MultUpdate's constructor applies for obj = :div (src/multupd.jl:37-40)."""
    X, W0, H0 = uniform(70, 90, 5, T, seed=5)
    with nmfx.Context(T, 70, 90, 5) as ctx:
        ctx.set_X(X)
        W, H = W0.copy(order="F"), H0.copy(order="F")
        res, trace = ctx.solve(nmfx._lib.ALG_MULTDIV, nmfx.make_opts(T, maxiter=12, tol=1e-30, track_objective=True), W, H)
    assert res.niters == 12 and np.isfinite(trace[:13]).all() and np.isfinite(W).all() and np.isfinite(H).all()
    lam = float(T(np.sqrt(np.finfo(T).eps)))
    Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
    ro = orc.solve("multdiv", X, Wc, Hc, orc.Opts(maxiter=12, tol=1e-30, lambda_w=lam, lambda_h=lam, track_objective=True))
    assert rel_trace_err(trace[:13], ro.trace) < TOL[T][0]


@pytest.mark.parametrize("shape", [(300, 260, 5), (1000, 1500, 64), (4096, 4096, 64), (700, 2100, 33)])
@pytest.mark.parametrize("update_H", [True, False])
def test_multmse_small_k_path(built, shape, update_H, monkeypatch):
    """k <= 64, Float32, P*N <= 4096^2: MultUpdate-MSE runs on the 4-launch stripe kernels (csrc/smallk.hpp) instead of the 12-launch
    general path.  Same algorithm, different summation order inside W'X / XH' / the Grams: the two device paths agree to f32 rounding
    (1e-5 on the objective trajectory, the tolerance this file states for f32 against the oracle), iteration counts are identical,
    and the oracle comparison holds for the new path on its own."""
    p, n, k = shape
    T = np.float32
    X, W0, H0 = planted(p, n, k, T, seed=p + k)
    lam = 1e-3
    alg = nmfx.MultUpdate(T, obj="mse", maxiter=12, tol=1e-30, lambda_w=lam, lambda_h=lam, update_H=update_H)
    runs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("NMFX_SMALLK", mode)
        W, H = W0.copy(order="F"), H0.copy(order="F")
        runs[mode] = (nmfx.solve(alg, X, W, H, track_objective=True), W, H)
    ra, rb = runs["1"][0], runs["0"][0]
    assert ra.niters == rb.niters == 12
    assert rel_trace_err(ra.trace, rb.trace) < 1e-5
    np.testing.assert_allclose(ra.info["relchange"][1:], rb.info["relchange"][1:], rtol=1e-3)
    assert np.max(np.abs(runs["1"][1] - runs["0"][1])) <= 1e-4 * np.max(np.abs(runs["0"][1]))
    assert np.max(np.abs(runs["1"][2] - runs["0"][2])) <= 1e-4 * np.max(np.abs(runs["0"][2]))
    if not update_H:
        assert np.array_equal(runs["1"][2], H0)
    if p * n <= 2_000_000:
        ro = orc.solve("multmse", X, W0.copy(order="F"), H0.copy(order="F"),
                       orc.Opts(maxiter=12, tol=1e-30, lambda_w=lam, lambda_h=lam, update_H=update_H, track_objective=True))
        assert rel_trace_err(ra.trace, ro.trace) < 1e-5


@pytest.mark.parametrize("shape", [(300, 260, 5), (1000, 1500, 64)])
@pytest.mark.parametrize("update_H", [True, False])
def test_small_k_path_stop_rule_in_the_finish_launch(built, shape, update_H):
    """Without objective tracking the small-k path evaluates stop_condition inside its W-side finish launch (the last of the 32
    statistics blocks to arrive; csrc/smallk.hpp), with tracking the separate check kernel does: both must stop at the same
    iteration with the same factors, for a tolerance that is reached and for one that is not."""
    p, n, k = shape
    T = np.float32
    X, W0, H0 = planted(p, n, k, T, seed=p + 3 * k)
    for tol, maxiter in ((2e-2, 200), (1e-30, 25)):
        out = []
        for track in (True, False):
            W, H = W0.copy(order="F"), H0.copy(order="F")
            r = nmfx.solve(nmfx.MultUpdate(T, obj="mse", maxiter=maxiter, tol=tol, update_H=update_H), X, W, H, track_objective=track)
            out.append((r, W, H))
        (ra, Wa, Ha), (rb, Wb, Hb) = out
        assert ra.niters == rb.niters and ra.converged == rb.converged
        assert (ra.converged and ra.niters < maxiter) if tol > 1e-10 else (not ra.converged and ra.niters == maxiter)
        assert np.array_equal(Wa, Wb) and np.array_equal(Ha, Hb)
        assert np.isclose(ra.objvalue, rb.objvalue, rtol=1e-6)


@pytest.mark.parametrize("T", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(300, 260, 5), (1000, 1500, 70), (513, 2100, 130)])
@pytest.mark.parametrize("update_H", [True, False])
def test_multdiv_fused_passes_are_bit_identical(built, T, shape, update_H, monkeypatch):
    """On one GPU multdiv's slab sum, scaling, stop_condition sums and the other side's divisor run as ONE pass per side
    (kernels.hpp: div_h_fused_kernel / div_w_fused_kernel) with the chunking and arithmetic of the separate kernels, which the
    multi-GPU paths still use (NMFX_DIV_FUSED=0 selects them here): every bit of W, H, the trace and the iteration count agrees."""
    p, n, k = shape
    X, W0, H0 = planted(p, n, k, T, seed=n + k)
    alg = nmfx.MultUpdate(T, obj="div", maxiter=9, tol=1e-30, update_H=update_H)
    runs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("NMFX_DIV_FUSED", mode)
        W, H = W0.copy(order="F"), H0.copy(order="F")
        runs[mode] = (nmfx.solve(alg, X, W, H, track_objective=True), W, H)
    ra, rb = runs["1"][0], runs["0"][0]
    assert ra.niters == rb.niters == 9
    assert np.array_equal(ra.trace, rb.trace)
    assert np.array_equal(runs["1"][1], runs["0"][1]) and np.array_equal(runs["1"][2], runs["0"][2])
    assert np.array_equal(ra.info["relchange"][1:], rb.info["relchange"][1:])


@pytest.mark.parametrize("shape", [(40, 50, 5), (100, 90, 64), (130, 170, 100), (200, 260, 256)])
@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_first_h_update_is_bit_identical_on_exact_inputs(built, T, shape):
    """MultUpdate-MSE with small-integer X, W0, H0: W'X, W'W and (W'W)H -- the device's association -- and the reference's W'(WH)
    are all exact in Float32, so the element-wise rule H .*= W'X ./ (W'WH .+ delta) (src/multupd.jl:98-106) receives the same
    numbers on both sides and H after the first iteration must agree BIT FOR BIT with the oracle (one rounded sum, one division,
    one product per element, in that order).  W is updated from that H, whose products are no longer exact: tolerance there."""
    p, n, k = shape
    rng = np.random.default_rng(7 + k)
    X = np.asfortranarray(rng.integers(0, 4, size=(p, n)).astype(T))
    W0 = np.asfortranarray(rng.integers(0, 3, size=(p, k)).astype(T))
    H0 = np.asfortranarray(rng.integers(0, 3, size=(k, n)).astype(T))
    assert 2 * 2 * k * p * 2 < 2 ** 24       # |(W'W)H| <= k * (p * 4) * 2
    o = nmfx.make_opts(T, maxiter=1, tol=1e-30, check_every=1000)
    Wg, Hg = W0.copy(order="F"), H0.copy(order="F")
    with nmfx.Context(T, p, n, k) as ctx:
        ctx.set_X(X)
        ctx.set_factors(Wg, Hg)
        ctx.iterate(0, o)
        ctx.get_factors(Wg, Hg)
    Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
    orc.solve("multmse", X, Wc, Hc, orc.Opts(maxiter=1, tol=1e-30))
    U = np.uint32 if T == np.float32 else np.uint64
    assert np.array_equal(Hg.view(U), Hc.view(U)), float(np.max(np.abs(Hg - Hc)))
    assert np.max(np.abs(Wg - Wc)) <= (1e-5 if T == np.float32 else 1e-13) * np.max(np.abs(Wc))


@pytest.mark.parametrize("shape", [(50, 40, 5), (90, 100, 64), (170, 130, 100), (260, 200, 256)])
@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_first_w_update_is_bit_identical_on_exact_inputs(built, T, shape):
    """The same for the W side (update_H = false, so W .*= XH' ./ (W(HH') .+ delta), src/multupd.jl:108-115, is the first thing
    that runs), and for the objective of the start point (0.5 * sqL2dist of integers: exact in Float64, src/multupd.jl:81)."""
    p, n, k = shape
    rng = np.random.default_rng(9 + k)
    X = np.asfortranarray(rng.integers(0, 4, size=(p, n)).astype(T))
    W0 = np.asfortranarray(rng.integers(0, 3, size=(p, k)).astype(T))
    H0 = np.asfortranarray(rng.integers(0, 3, size=(k, n)).astype(T))
    assert 2 * 2 * k * n * 2 < 2 ** 24
    o = nmfx.make_opts(T, maxiter=1, tol=1e-30, update_H=False, check_every=1000)
    Wg, Hg = W0.copy(order="F"), H0.copy(order="F")
    with nmfx.Context(T, p, n, k) as ctx:
        ctx.set_X(X)
        ctx.set_factors(Wg, Hg)
        obj0 = ctx.objective(0, o)
        ctx.iterate(0, o)
        ctx.get_factors(Wg, Hg)
    Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
    orc.solve("multmse", X, Wc, Hc, orc.Opts(maxiter=1, tol=1e-30, update_H=False))
    U = np.uint32 if T == np.float32 else np.uint64
    assert np.array_equal(Wg.view(U), Wc.view(U)), float(np.max(np.abs(Wg - Wc)))
    assert np.array_equal(Hg, H0)
    R = X.astype(np.float64) - W0.astype(np.float64) @ H0.astype(np.float64)
    assert obj0 == float(T(0.5 * np.sum(R * R)))


@pytest.mark.parametrize("shape", [(40, 50, 5), (200, 90, 64), (300, 170, 100), (800, 260, 256), (8192, 8192, 128)])
@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_multdiv_first_h_update_is_bit_identical_on_selection_factors(built, T, shape):
    """MultUpdate-div (src/multupd.jl:172-179): Q = X ./ (WH .+ delta), H .*= (W'Q) ./ (sum(W, dims=1)' .+ lambda).  With a W0
    whose column j is the indicator of ONE row r_j, WH is exact (row r_j of it is row j of H0), so Q is the same rounded quotient
    on both sides, W'Q selects row r_j of Q (sums with a single non-zero term: exact in any order) and the column sums are 1:
    H after the first iteration must agree BIT FOR BIT with the oracle -- a test of the ratio epilogue's division (the IEEE
    sequence written out in pieces inside the persistent W*H kernel at the last shape, the block-per-tile kernel before) and of
    the scaling pass, on non-trivial quotients."""
    p, n, k = shape
    if T == np.float64 and p > 4096:
        pytest.skip("the persistent W*H kernel is Float32 only")
    rng = np.random.default_rng(13 + k)
    X = np.asfortranarray(rng.integers(0, 50, size=(p, n)).astype(T))
    rows = rng.permutation(p)[:k]
    W0 = np.zeros((p, k), dtype=T, order="F")
    W0[rows, np.arange(k)] = 1
    H0 = np.asfortranarray(rng.integers(0, 7, size=(k, n)).astype(T))
    lam = 0.25
    o = nmfx.make_opts(T, maxiter=1, tol=1e-30, lambda_w=lam, lambda_h=lam, check_every=1000)
    Wg, Hg = W0.copy(order="F"), H0.copy(order="F")
    with nmfx.Context(T, p, n, k) as ctx:
        ctx.set_X(X)
        ctx.set_factors(Wg, Hg)
        ctx.iterate(1, o)
        ctx.get_factors(Wg, Hg)
    Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
    orc.solve("multdiv", X, Wc, Hc, orc.Opts(maxiter=1, tol=1e-30, lambda_w=lam, lambda_h=lam))
    U = np.uint32 if T == np.float32 else np.uint64
    assert np.array_equal(Hg.view(U), Hc.view(U)), float(np.max(np.abs(Hg - Hc)))
    assert not np.array_equal(Hg, H0)


@pytest.mark.parametrize("shape", [(300, 260, 70), (2048, 2304, 128), (129, 257, 128)])
def test_ratio_pass_division_against_the_ieee_sequence(built, shape, monkeypatch):
    """Round 6: the ratio pass Q = X ./ (WH + delta) of MultUpdate(:div) (src/multupd.jl:172-174, 184-186) divides by v_rcp_f32's
    reciprocal and one residual correction formed exactly by a fused multiply-add (gemm_mfma.hpp: ratio_div_fast; 4 instructions
    for the 12 of the IEEE sequence in the matrix cores' shadow).  Emulated on the host over 2e7 operand pairs the result equals the
    correctly rounded quotient everywhere when the reciprocal is correctly rounded and differs by ONE ulp on 2e-7 of the pairs when
    the reciprocal is an ulp off.  Here: ten Float32 iterations with either division (NMFX_DIV_IEEE=1 keeps the IEEE sequence; both
    shapes of the product kernel: block-per-tile and the persistent stream form at 2048 x 2304, k = 128) agree to rounding -- far
    inside the 1e-5 of the objective's stated tolerance -- and with the oracle, whose division is IEEE."""
    p, n, k = shape
    T = np.float32
    X, W0, H0 = planted(p, n, k, T, seed=31 + p)
    alg = nmfx.MultUpdate(T, obj="div", maxiter=10, tol=1e-30)
    out = {}
    for ieee in ("0", "1"):
        monkeypatch.setenv("NMFX_DIV_IEEE", ieee)
        W, H = W0.copy(order="F"), H0.copy(order="F")
        r = nmfx.solve(alg, X, W, H, track_objective=True)
        out[ieee] = (r, W, H)
    monkeypatch.delenv("NMFX_DIV_IEEE")
    (ra, Wa, Ha), (rb, Wb, Hb) = out["0"], out["1"]
    assert ra.niters == rb.niters == 10
    assert rel_trace_err(ra.trace, rb.trace) < 2e-6
    assert np.max(np.abs(Wa - Wb)) <= 2e-5 * np.max(np.abs(Wb)) and np.max(np.abs(Ha - Hb)) <= 2e-5 * np.max(np.abs(Hb))
    ro = orc.solve("multdiv", X, W0.copy(order="F"), H0.copy(order="F"), orc.Opts(maxiter=10, tol=1e-30, track_objective=True))
    assert rel_trace_err(ra.trace, ro.trace) < TOL[T][0]
