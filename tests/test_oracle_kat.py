"""Pin the oracle: every known-answer test the reference's own test-suite holds for the hot path,
run against BOTH restatements (NumPy twin and C), plus their mutual agreement.

Reference tests ported (file:line under /root/reference/test):
  testproblems.jl:6-13   laurberg6x3
  multupd.jl:3-22        MultUpdate KAT (both dtypes, both objectives, lambda in {0, 1e-4})
  alspgrad.jl:3-25       sub-solver KATs + solve! smoke
  utils.jl:6-15,29-34,48-63   adddiag!, projectnn!, pdsolve!, pdrsolve!
  interf.jl:33-37        update_H=false leaves H untouched
  coorddesc.jl:5-14      CoordinateDescent KATs (shuffle = true with the component orders as an input: Julia's RNG is not available)
  greedycd.jl:5-20       GreedyCD KATs (lambda_w, lambda_h in {0, 1e-5})
  spa.jl:11-32           spa(X, k): non-negative factors, reconstruction of a rank-k product and of separable data
"""
import numpy as np
import pytest

import c_oracle as co
import nmf_oracle as orc
from problems import planted, rel_trace_err

ORACLES = {"numpy": orc, "c": co}


def test_laurberg_problem():
    X, W, H = orc.laurberg6x3(0.3)
    assert X.shape == (6, 6) and W.shape == (6, 3) and H.shape == (3, 6)
    assert np.array_equal(W, H.T) and np.allclose(X, W @ H)
    assert H[0, 0] == 0.3 and H[0, 1] == 1 and H[2, 5] == 0.3 and H[1, 2] == 0


@pytest.mark.parametrize("impl", ["numpy", "c"])
@pytest.mark.parametrize("T", [np.float64, np.float32])
@pytest.mark.parametrize("alg", ["multmse", "multdiv"])
@pytest.mark.parametrize("lam", [0.0, 1e-4])
def test_multupd_kat(impl, T, alg, lam):
    X, Wg, Hg = orc.laurberg6x3(0.3, T)
    rng = np.random.default_rng(0)
    W = np.asfortranarray(Wg + rng.random(Wg.shape).astype(T) * T(0.1))
    H = Hg.copy(order="F")
    ORACLES[impl].solve(alg, X, W, H, orc.Opts(maxiter=5000, tol=1e-9, lambda_w=lam, lambda_h=lam))
    assert np.all(W >= 0) and np.all(H >= 0)
    assert not np.isnan(W).any() and not np.isnan(H).any()
    assert np.linalg.norm(X - W @ H) <= 1e-2          # X ≈ W*Hg atol=1e-2


@pytest.mark.parametrize("impl", ["numpy", "c"])
@pytest.mark.parametrize("T", [np.float64, np.float32])
def test_alspgrad_subsolver_kat(impl, T):
    o = ORACLES[impl]
    X, Wg, Hg = orc.laurberg6x3(0.3, T)
    rng = np.random.default_rng(1)
    eps = np.finfo(T).eps
    H = np.asfortranarray(rng.random(Hg.shape).astype(T))
    o.alspgrad_updateh(X, Wg, H, maxiter=1000, tolg=eps)
    assert np.all(H >= 0) and np.linalg.norm(H - Hg) <= eps ** 0.25
    W = np.asfortranarray(rng.random(Wg.shape).astype(T))
    o.alspgrad_updatew(X, W, Hg, maxiter=1000, tolg=eps)
    assert np.all(W >= 0) and np.linalg.norm(W - Wg) <= eps ** 0.25
    r = o.solve("alspgrad", X, W, H)                # smoke: NMF.solve!(NMF.ALSPGrad{T}(), X, W, H)
    assert r.niters >= 1 and np.isfinite(r.objvalue)


@pytest.mark.parametrize("impl", ["numpy", "c"])
@pytest.mark.parametrize("T", [np.float64, np.float32])
def test_coorddesc_kat(impl, T):
    """test/coorddesc.jl:5-14."""
    rng = np.random.default_rng(2)
    X, Wg, Hg = orc.laurberg6x3(0.3, T)
    W = np.asfortranarray(Wg + rng.random(Wg.shape).astype(T) * T(0.1)); H = Hg.copy(order="F")
    ORACLES[impl].solve("cd", X, W, H, orc.Opts(maxiter=1000, tol=1e-9))
    assert np.allclose(X, W @ H, atol=1e-4, rtol=0)
    W = np.asfortranarray(Wg + rng.random(Wg.shape).astype(T) * T(0.1)); H = Hg.copy(order="F")
    # alpha = 1e-4, l1ratio = 0.5, regularization = :both  ->  l1 = l2 = 5e-5 on both sides (coorddesc.jl:62-82)
    ORACLES[impl].solve("cd", X, W, H, orc.Opts(maxiter=1000, tol=1e-9, l1_w=5e-5, l2_w=5e-5, l1_h=5e-5, l2_h=5e-5))
    assert np.allclose(X, W @ H, atol=1e-2, rtol=0)


@pytest.mark.parametrize("T", [np.float64, np.float32])
def test_coorddesc_shuffle_kat(T):
    """test/coorddesc.jl:10-14 with shuffle = true: the reference draws randperm(k) per _update_coord_descent! call from Julia's
    RNG; the oracle takes the orders as an input (here the documented Philox orders the device uses), and the KAT -- the
    regularised solve still reconstructs X to 1e-2 -- holds for them as for any other sequence of orders."""
    import philox_ref
    rng = np.random.default_rng(2)
    X, Wg, Hg = orc.laurberg6x3(0.3, T)
    W = np.asfortranarray(Wg + rng.random(Wg.shape).astype(T) * T(0.1)); H = Hg.copy(order="F")
    o = orc.Opts(maxiter=1000, tol=1e-9, l1_w=5e-5, l2_w=5e-5, l1_h=5e-5, l2_h=5e-5, perm_source=lambda c: philox_ref.cd_permutation(3, 7, c))
    r = orc.solve("cd", X, W, H, o)
    assert np.allclose(X, W @ H, atol=1e-2, rtol=0)
    # the orders matter: a different key gives a different trajectory; shuffle = false a third one
    W2 = np.asfortranarray(Wg + np.random.default_rng(2).random(Wg.shape).astype(T) * T(0.1)); H2 = Hg.copy(order="F")
    r2 = orc.solve("cd", X, W2, H2, orc.Opts(maxiter=1000, tol=1e-9, l1_w=5e-5, l2_w=5e-5, l1_h=5e-5, l2_h=5e-5))
    assert not np.array_equal(W, W2)
    for c in range(6):                                   # every order is a permutation of the components
        assert sorted(philox_ref.cd_permutation(5, 11, c)) == list(range(5))
    assert any(list(philox_ref.cd_permutation(5, 11, c)) != list(range(5)) for c in range(6))


@pytest.mark.parametrize("impl", ["numpy", "c"])
@pytest.mark.parametrize("T", [np.float64, np.float32])
@pytest.mark.parametrize("lw", [0.0, 1e-5])
@pytest.mark.parametrize("lh", [0.0, 1e-5])
def test_greedycd_kat(impl, T, lw, lh):
    """test/greedycd.jl:5-20."""
    rng = np.random.default_rng(6)
    X, Wg, Hg = orc.laurberg6x3(0.3, T)
    W = np.asfortranarray(Wg + rng.random(Wg.shape).astype(T) * T(0.1)); H = Hg.copy(order="F")
    ORACLES[impl].solve("greedycd", X, W, H, orc.Opts(maxiter=1000, tol=1e-9, lambda_w=lw, lambda_h=lh))
    assert np.all(W >= 0) and np.all(H >= 0) and not np.isnan(W).any() and not np.isnan(H).any()
    assert np.allclose(X, W @ H, atol=1e-3, rtol=0)


def _pdmat(rng, n):
    g = rng.standard_normal((n, n))
    return np.asfortranarray(g.T @ g + 0.1 * np.eye(n))


def test_utils_kat_numpy():
    rng = np.random.default_rng(2)
    a0 = rng.random((3, 3))
    a = a0.copy()
    orc.adddiag(a, 0.0)
    assert np.array_equal(a, a0)
    orc.adddiag(a, 2.5)
    assert np.array_equal(a, a0 + 2.5 * np.eye(3))
    b0 = rng.standard_normal((5, 5))
    b = b0.copy()
    orc.projectnn(b)
    assert np.array_equal(b, np.maximum(b0, 0.0))
    A = _pdmat(rng, 5)
    Xs = rng.random((5, 3))
    assert np.allclose(orc.pdsolve(A, np.asfortranarray(A @ Xs)), Xs)
    B = _pdmat(rng, 5)
    Xr = rng.random((4, 5))
    assert np.allclose(orc.pdrsolve(np.asfortranarray(Xr @ B), B), Xr)


def test_utils_kat_c():
    rng = np.random.default_rng(3)
    for T, tol in ((np.float64, 1e-9), (np.float32, 2e-3)):
        A = _pdmat(rng, 5).astype(T)
        Xs = rng.random((5, 3)).astype(T)
        assert np.allclose(co.pdsolve(A, A @ Xs), Xs, atol=tol)
        B = _pdmat(rng, 5).astype(T)
        Xr = rng.random((4, 5)).astype(T)
        assert np.allclose(co.pdrsolve(Xr @ B, B), Xr, atol=tol)
    with pytest.raises(np.linalg.LinAlgError):
        co.pdsolve(-np.eye(3), np.ones((3, 1)))
    with pytest.raises(orc.PosDefException):
        orc.pdsolve(np.asfortranarray(-np.eye(3)), np.ones((3, 1), order="F"))


@pytest.mark.parametrize("impl", ["numpy", "c"])
@pytest.mark.parametrize("alg", ["multmse", "multdiv", "projals", "alspgrad", "cd", "greedycd"])
def test_update_H_false_keeps_H(impl, alg):
    rng = np.random.default_rng(4)
    T = np.float64
    Wg = np.maximum(rng.random((5, 3)) - 0.3, 0)
    Hg = np.maximum(rng.random((3, 8)) - 0.3, 0)
    X = np.asfortranarray(Wg @ Hg)
    W = np.asfortranarray(np.maximum(rng.random((5, 3)) - 0.3, 0))
    H = np.asfortranarray(np.maximum(rng.random((3, 8)) - 0.3, 0))
    W0, H0 = W.copy(), H.copy()
    ORACLES[impl].solve(alg, X, W, H, orc.Opts(update_H=False, tol=float(np.cbrt(np.finfo(T).eps / 100))))
    assert np.array_equal(H, H0) and np.any(W != W0)


def test_stop_condition_agrees():
    rng = np.random.default_rng(5)
    for T in (np.float32, np.float64):
        W = np.asfortranarray(rng.random((30, 4)).astype(T))
        H = np.asfortranarray(rng.random((4, 50)).astype(T))
        for scale, expect in ((1e-1, False), (1e-9, True)):
            pW = np.asfortranarray(W * T(1 + scale))
            pH = np.asfortranarray(H * T(1 - scale))
            tol = float(T(1e-3))
            assert orc.stop_condition(W, pW, H, pH, tol) == expect
            assert co.stop_condition(W, pW, H, pH, tol) == expect


@pytest.mark.parametrize("T,lim", [(np.float32, 2e-6), (np.float64, 1e-14)])
def test_preallocated_multmse_state_is_the_same_iteration(T, lim):
    """bench.py's cpu_baseline times _MultMSEState (prepare_state's arrays allocated once, src/multupd.jl:63-80, every mul! and
    loop in place): the same operations in the same order as the allocating _MultMSE -- factors equal to a few ulp, the stop
    decision identical."""
    X, W0, H0 = planted(70, 90, 6, T, seed=3)
    o = orc.resolve_opts(orc.MULTMSE, T, orc.Opts(maxiter=3, tol=1e-30, lambda_w=1e-3, lambda_h=2e-3))
    W1, H1, W2, H2 = (a.copy(order="F") for a in (W0, H0, W0, H0))
    a, b = orc._MultMSE(T, o, X, W1, H1), orc._MultMSEState(T, o, X, W2, H2)
    for tol in (1e-30, 1e-30, 0.5, 1e-30):
        pW, pH = W1.copy(), H1.copy()
        a.update(X, W1, H1)
        np.copyto(b.preW, W2)
        np.copyto(b.preH, H2)
        phases = {}
        b.update(X, W2, H2, phases)
        assert len(phases) == 8
        assert orc.stop_condition(W1, pW, H1, pH, T(tol)) == b.stop_condition(W2, H2, tol)
        assert np.abs(W1 - W2).max() <= lim * np.abs(W1).max() and np.abs(H1 - H2).max() <= lim * np.abs(H1).max()


@pytest.mark.parametrize("T,lims", [(np.float64, (1e-12, 1e-12, 1e-9, 1e-9, 1e-12, 1e-11)), (np.float32, (2e-6, 2e-6, 2e-3, 1e-3, 1e-4, 2e-3))])
def test_two_restatements_agree(T, lims):
    """The C and NumPy restatements are independent (own GEMM / Cholesky / reductions vs BLAS / LAPACK)."""
    for alg, lim in zip(("multmse", "multdiv", "projals", "alspgrad", "cd", "greedycd"), lims):
        X, W0, H0 = planted(33, 47, 4, T, seed=77, normalize=(alg != "projals"))
        extra = dict(lambda_w=0.05, lambda_h=0.05) if alg == "projals" else {}
        o = orc.Opts(maxiter=15, tol=1e-30, track_objective=True, **extra)
        a = orc.solve(alg, X, W0.copy(order="F"), H0.copy(order="F"), o)
        b = co.solve(alg, X, W0.copy(order="F"), H0.copy(order="F"), o)
        assert a.niters == b.niters
        assert rel_trace_err(a.trace, b.trace) < lim, alg


def test_defaults_match_reference_table():
    """dtype-dependent defaults."""
    o32 = orc.resolve_opts(orc.MULTMSE, np.float32, orc.Opts())
    assert abs(o32.tol - 4.92e-3) < 1e-5 and abs(o32.delta - 3.4527e-4) < 1e-8
    o64 = orc.resolve_opts(orc.MULTDIV, np.float64, orc.Opts())
    assert abs(o64.tol - 6.06e-6) < 1e-8 and abs(o64.lambda_w - 1.4901e-8) < 1e-12 and o64.lambda_h == o64.lambda_w
    oa = orc.resolve_opts(orc.ALSPGRAD, np.float32, orc.Opts())
    assert abs(oa.tolg - 1.858e-2) < 1e-5 and oa.maxsubiter == 200 and oa.traceiter == 20
    op = orc.resolve_opts(orc.PROJALS, np.float64, orc.Opts())
    assert abs(op.lambda_w - 6.06e-6) < 1e-8


def separable_data(m, n, k, rng, T):
    """separable_data(m, n, k) of src/spa.jl:28-36 (own RNG): W = rand, H = [I V] with the columns of V summing to one, columns
    of H permuted -- so the (scaled) columns of W appear among the columns of X = W H."""
    W = rng.random((m, k))
    V = rng.random((k, n - k))
    V /= V.sum(axis=0, keepdims=True)
    H = np.concatenate([np.eye(k), V], axis=1)[:, rng.permutation(n)]
    return W.astype(T), H.astype(T)


@pytest.mark.parametrize("T", [np.float64, np.float32])
def test_spa_kat(T):
    """test/spa.jl:11-32."""
    rng = np.random.default_rng(4)
    p, n, k = 15, 8, 2
    eps4 = np.finfo(T).eps ** 0.25
    Wg = np.maximum(rng.random((p, k)).astype(T) - T(0.3), T(eps4))
    Hg = np.maximum(rng.random((k, n)).astype(T) - T(0.3), T(eps4))
    X = np.asfortranarray(Wg @ Hg)
    w, h, ai = orc.spa(X, k)
    assert np.all(w >= 0) and np.all(h >= 0) and len(set(ai)) == k
    assert np.allclose(w @ h, X, atol=10.0 * eps4, rtol=0)
    Wg, Hg = separable_data(p, n, k, rng, T)
    X = np.asfortranarray(Wg @ Hg)
    w, h, ai = orc.spa(X, k)
    assert np.all(w >= 0) and np.all(h >= 0)
    d = (X - w @ h).astype(np.float64)
    assert float(np.sum(d * d)) < np.finfo(T).eps                      # sqL2dist(X, x) < eps(T)
    # the anchors are the columns where H is a unit vector
    assert sorted(ai) == sorted(int(np.argmax(Hg[a] == 1)) for a in range(k))
    r = orc.spa_solve(X, w, h)
    assert r.niters == 0 and r.converged and r.objvalue < np.finfo(T).eps
