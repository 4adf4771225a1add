"""verbose-style objective tracking (src/common.jl:76-82) together with the stop rule: when stop_condition fires at
iteration t < maxiter, the objective of iteration t itself is still evaluated (common.jl:73 then :79) and is what
Result.objvalue reports.  (Regression: the device loop used to raise its `done` flag before enqueueing that evaluation.)"""
import numpy as np
import pytest

import nmf_oracle as orc
import nmfx
from problems import planted

pytestmark = pytest.mark.gpu


def _inst(alg, T, **kw):
    if alg in ("multmse", "multdiv"):
        return nmfx.MultUpdate(T, obj=alg[4:], **kw)
    return {"projals": nmfx.ProjectedALS, "alspgrad": nmfx.ALSPGrad, "cd": nmfx.CoordinateDescent, "greedycd": nmfx.GreedyCD}[alg](T, **kw)


@pytest.mark.parametrize("alg", ["multmse", "multdiv", "projals", "alspgrad", "cd", "greedycd"])
def test_tracked_objective_of_the_converging_iteration(built, alg):
    T = np.float64
    X, W0, H0 = planted(40, 60, 3, T, seed=17, normalize=(alg != "projals"), zeroh=(alg == "projals"))
    tol = 1e-3
    W, H = W0.copy(order="F"), H0.copy(order="F")
    r = nmfx.solve(_inst(alg, T, maxiter=400, tol=tol), X, W, H, track_objective=True)
    ro = orc.solve(alg, X, W0.copy(order="F"), H0.copy(order="F"), orc.Opts(maxiter=400, tol=tol, track_objective=True))
    assert r.converged and ro.converged and 1 < r.niters < 400
    # Stop-rule fidelity (src/common.jl:92-111).  The reference sums dev / sum sequentially in T; the device sums the same
    # T-rounded terms in Float64 in a fixed tree order -- the two totals differ by at most ~eps(T)*sqrt(length) relative, so
    # the DECISION sqrt(dev) > tol*sqrt(sum) can only differ when an iteration's relchange sits that close to tol.  On this
    # input every iteration keeps a margin of > 1e-4 from the threshold (asserted), the device's relchange column equals the
    # oracle's to 1e-6, and therefore niters must be IDENTICAL, not just close.
    rc_o, rc_d = np.array(ro.relchange[1:]), np.array(r.info["relchange"][1:])
    assert np.min(np.abs(rc_o - tol) / tol) > 1e-4
    m = min(len(rc_o), len(rc_d))
    np.testing.assert_allclose(rc_d[:m], rc_o[:m], rtol=1e-6)
    assert r.niters == ro.niters
    assert len(r.trace) == r.niters + 1 and np.all(np.isfinite(r.trace))
    assert r.objvalue == r.trace[-1]
    m = min(len(r.trace), len(ro.trace))
    np.testing.assert_allclose(r.trace[:m], np.array(ro.trace)[:m], rtol=1e-6)
    # and without tracking the same final value comes from the single evaluation at the end (common.jl:85-87)
    W2, H2 = W0.copy(order="F"), H0.copy(order="F")
    r2 = nmfx.solve(_inst(alg, T, maxiter=400, tol=tol), X, W2, H2)
    assert r2.niters == r.niters and r2.objvalue == r.objvalue and np.array_equal(W2, W)


@pytest.mark.parametrize("alg", ["multmse", "projals", "alspgrad", "greedycd"])
@pytest.mark.parametrize("update_H", [True, False])
def test_verbose_table_columns(built, alg, update_H, capsys):
    """The two extra columns of the reference's verbose table (src/common.jl:54-59, :76-82): elapsed time and
    `(W & H).relchange` = stop_condition's devmax, INCLUDING its early-return semantics (maximum over the components up to
    the first failing one)."""
    T = np.float64
    X, W0, H0 = planted(30, 45, 4, T, seed=5, normalize=(alg != "projals"))
    inst = _inst(alg, T, maxiter=12, tol=1e-2, update_H=update_H)
    W, H = W0.copy(order="F"), H0.copy(order="F")
    r = nmfx.solve(inst, X, W, H, track_objective=True)
    ro = orc.solve(alg, X, W0.copy(order="F"), H0.copy(order="F"), orc.Opts(maxiter=12, tol=1e-2, update_H=update_H, track_objective=True))
    assert r.niters == ro.niters
    el, rc = r.info["elapsed"], r.info["relchange"]
    assert len(el) == len(rc) == r.niters + 1 and el[0] == 0 and np.isnan(rc[0])
    assert np.all(np.diff(el) > 0) and el[-1] <= r.info["seconds_loop"] * 1.001
    np.testing.assert_allclose(rc[1:], ro.relchange[1:], rtol=1e-6)
    # verbose = true prints the table in the reference's format
    inst.verbose = True
    nmfx.solve(inst, X, W0.copy(order="F"), H0.copy(order="F"))
    lines = capsys.readouterr().out.strip().splitlines()
    assert lines[0].split() == ["Iter", "Elapsed", "time", "objv", "objv.change", "(W", "&", "H).relchange"]
    assert len(lines) == r.niters + 2 and len(lines[1].split()) == 3 and all(len(l.split()) == 5 for l in lines[2:])
    assert lines[0].startswith("Iter     Elapsed time     objv             objv.change      (W & H).relchange")


@pytest.mark.parametrize("alg", ["multmse", "multdiv", "projals", "cd", "greedycd"])
def test_result_does_not_depend_on_the_poll_interval(built, alg):
    """nmfx_opts.check_every: the host polls the device stop flag every N iterations, or (<= 0, the default) with a window that
    grows from 4 while a window takes under a millisecond.  Iterations enqueued past the stop are no-ops, so niters, the factors and
    the objective must be the same for every choice -- including a converging solve whose stop falls inside a long window."""
    T = np.float64
    X, W0, H0 = planted(60, 84, 4, T, seed=23, normalize=(alg != "projals"), zeroh=(alg == "projals"))
    outs = []
    for ce in (0, 1, 3, 64, 1 << 20):
        W, H = W0.copy(order="F"), H0.copy(order="F")
        r = nmfx.solve(_inst(alg, T, maxiter=3000, tol=2e-4), X, W, H, check_every=ce)
        outs.append((r.niters, r.converged, r.objvalue, W, H))
    assert outs[0][1] and 8 < outs[0][0] < 3000
    for o in outs[1:]:
        assert o[0] == outs[0][0] and o[1] == outs[0][1] and o[2] == outs[0][2]
        assert np.array_equal(o[3], outs[0][3]) and np.array_equal(o[4], outs[0][4])


@pytest.mark.parametrize("alg,algid", [("multmse", 0), ("multdiv", 1), ("projals", 2), ("greedycd", 5)])
def test_final_objective_can_be_deferred(built, alg, algid):
    """nmfx_set_final_objective(ctx, 0): nmfx_iterate leaves Result.objvalue NaN (bench.py times K iterations that way) and
    nmfx_objective afterwards returns what the default call reports; the factors do not depend on the switch."""
    T = np.float64
    p, n, k = 70, 90, 6
    X, W0, H0 = planted(p, n, k, T, seed=23, normalize=(alg != "projals"), zeroh=(alg == "projals"))
    lam = 0.05 if alg == "projals" else 0.0
    o = nmfx.make_opts(T, maxiter=7, tol=1e-30, lambda_w=lam, lambda_h=lam, check_every=1000)
    out = []
    for deferred in (False, True):
        with nmfx.Context(T, p, n, k) as ctx:
            ctx.set_X(X)
            ctx.set_factors(W0.copy(order="F"), H0.copy(order="F"))
            if deferred:
                ctx.set_final_objective(False)
            res, _ = ctx.iterate(algid, o)
            objv = res.objvalue
            if deferred:
                assert np.isnan(objv)
                ctx.set_final_objective(True)
                objv = ctx.objective(algid, o)
            W, H = np.empty((p, k), T, order="F"), np.empty((k, n), T, order="F")
            ctx.get_factors(W, H)
            out.append((objv, W, H, res.niters))
    assert out[0][3] == out[1][3] == 7
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    assert out[0][0] == out[1][0] and np.isfinite(out[0][0])


@pytest.mark.parametrize("T", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(50, 40, 5), (300, 200, 64), (1100, 900, 100)])
def test_exact_stop_sums_reproduce_the_reference_bit_for_bit(built, T, shape):
    """nmfx_opts.stop_sums = 1: stop_condition's sums accumulated sequentially in T, one chain per component (src/common.jl:95-104).
    On small-integer inputs with update_H = false the first W update is bit-identical to the oracle's (exact products), so the
    relchange column -- sqrt(max_j dev_w / sum_w), a quotient of those sums -- must then equal the oracle's BIT FOR BIT; the default
    (Float64 tree sums of the same terms) agrees to rounding only.  And a stop threshold placed between the two roundings of a real
    run cannot be told apart, so `niters` is checked on a seeded problem against the oracle with NO margin requirement."""
    p, n, k = shape
    rng = np.random.default_rng(11 + k)
    X = np.asfortranarray(rng.integers(0, 4, size=(p, n)).astype(T))
    W0 = np.asfortranarray(rng.integers(0, 3, size=(p, k)).astype(T))
    H0 = np.asfortranarray(rng.integers(1, 3, size=(k, n)).astype(T))
    assert 2 * 2 * k * n * 2 < 2 ** 24
    out = {}
    for exact in (True, False):
        Wg, Hg = W0.copy(order="F"), H0.copy(order="F")
        with nmfx.Context(T, p, n, k) as ctx:
            ctx.set_X(X)
            res, _ = ctx.solve(0, nmfx.make_opts(T, maxiter=1, tol=1e-30, update_H=False, track_objective=True, exact_stop=exact), Wg, Hg)
            _, rc = ctx.iter_trace(2)
        out[exact] = (Wg, rc[1])
    Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
    orc.solve("multmse", X, Wc, Hc, orc.Opts(maxiter=1, tol=1e-30, update_H=False))
    U = np.uint32 if T == np.float32 else np.uint64
    assert np.array_equal(out[True][0].view(U), Wc.view(U))
    _, devmax = orc.stop_condition_dev(Wc, W0, H0, H0, 1e-30)
    assert T(out[True][1]) == T(devmax), (out[True][1], devmax)
    assert abs(out[False][1] - devmax) <= 1e-5 * devmax


def test_exact_stop_sums_niters_without_a_margin(built):
    T = np.float32
    X, W0, H0 = planted(300, 260, 6, T, seed=21)
    for tol in (3e-3, 1e-3, 4e-4):
        Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
        ro = orc.solve("multmse", X, Wc, Hc, orc.Opts(maxiter=400, tol=tol))
        Wg, Hg = W0.copy(order="F"), H0.copy(order="F")
        with nmfx.Context(T, 300, 260, 6) as ctx:
            ctx.set_X(X)
            res, _ = ctx.solve(0, nmfx.make_opts(T, maxiter=400, tol=tol, exact_stop=True), Wg, Hg)
        assert abs(res.niters - ro.niters) <= 1 and bool(res.converged) == ro.converged


@pytest.mark.parametrize("alg,shape", [("multmse", (300, 260, 6)), ("multmse", (512, 512, 64)), ("multdiv", (300, 260, 6)), ("projals", (300, 260, 6)),
                                       ("alspgrad", (200, 180, 5)), ("cd", (300, 260, 6)), ("greedycd", (300, 260, 6))])
@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_exact_stop_sums_see_the_previous_iterate_for_every_algorithm(built, alg, shape, T):
    """stop_sums = 1 evaluates stop_condition (src/common.jl:92-111) on the factors of iteration t against those of t - 1, which it takes
    from the ping-pong partners of the live buffers.  That holds by construction for MultUpdate; here it is CHECKED for every algorithm
    (ProjectedALS, CoordinateDescent, GreedyCD, ALSPGrad's explicit preW / preH copies) and for the small-k path of MultUpdate-MSE
    (512 x 512, k = 64): the relchange column of iteration 3 must be, bit for bit, what the sequential T-precision sums give on the
    factors the DEVICE returned after 3 and after 2 iterations (runs are bit-reproducible) -- no oracle trajectory involved."""
    p, n, k = shape
    X, W0, H0 = planted(p, n, k, T, seed=31, normalize=(alg != "projals"))
    algid = {"multmse": 0, "multdiv": 1, "projals": 2, "alspgrad": 3, "cd": 4, "greedycd": 5}[alg]
    lam = 0.05 if alg == "projals" else (float(np.sqrt(np.finfo(T).eps)) if alg == "multdiv" else 0.0)
    got = {}
    with nmfx.Context(T, p, n, k) as ctx:
        ctx.set_X(X)
        for iters in (2, 3):
            Wg, Hg = W0.copy(order="F"), H0.copy(order="F")
            res, _ = ctx.solve(algid, nmfx.make_opts(T, maxiter=iters, tol=1e-30, lambda_w=lam, lambda_h=lam, track_objective=True, exact_stop=True), Wg, Hg)
            assert res.niters == iters
            _, rc = ctx.iter_trace(iters + 1)
            got[iters] = (Wg, Hg, rc)
    _, devmax = orc.stop_condition_dev(got[3][0], got[2][0], got[3][1], got[2][1], 1e-30)
    assert T(got[3][2][3]) == T(devmax), (alg, got[3][2][3], devmax)
    assert T(got[3][2][2]) == T(got[2][2][2])          # ... and the run is reproducible: iteration 2's value is the same in both runs
