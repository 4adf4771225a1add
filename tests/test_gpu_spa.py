"""GPU parity: spa(X, k) / nnmf(init = :spa) / nnmf(alg = :spa) through the C ABI (src/spa.jl, src/interf.jl:50-51, 73-77).

Stated tolerances.  Anchors: identical indices to the oracle's (the inputs keep the arg-max margins far above rounding).
W = X[:, anchors]: bit-exact.  H: the reference takes the NNLS minimiser from NonNegLeastSquares.fnnls; the device runs the same
active-set method (warm-started), the oracle uses SciPy's Lawson-Hanson solver on W, x directly: same minimiser, different
arithmetic (normal equations vs QR).  Entrywise: 2e-6 (f64) / 5e-3 (f32) of max|H| -- H inherits cond(W'W) eps(T), cond up to
4.5e7 for the square-W shape, where Float32 normal equations carry no digits and only the residual is compared.  Residual
||X - WH||, which is what NNLS minimises: within 1e-9 (f64) / 1e-3 (f32) of ||X|| of the oracle's -- normal equations formed in T
resolve the residual to sqrt(eps(T)) ||X|| (3.4e-4 in f32; measured 1.7e-4, scripts/spa_diag.py), for the reference's fnnls too,
which receives W'W and W'X in T."""
import numpy as np
import pytest

import nmf_oracle as orc
import nmfx
from test_oracle_kat import separable_data

pytestmark = pytest.mark.gpu
HTOL = {np.float64: 2e-6, np.float32: 5e-3}
RTOL = {np.float64: 1e-9, np.float32: 1e-3}


@pytest.mark.parametrize("T", [np.float64, np.float32])
def test_spa_reference_kat(built, T):
    """test/spa.jl:11-32 on the device: exact recovery of a separable X (up to the anchor order)."""
    rng = np.random.default_rng(4)
    p, n, k = 15, 8, 2
    eps4 = np.finfo(T).eps ** 0.25
    Wg = np.maximum(rng.random((p, k)).astype(T) - T(0.3), T(eps4))
    Hg = np.maximum(rng.random((k, n)).astype(T) - T(0.3), T(eps4))
    X = np.asfortranarray(Wg @ Hg)
    w, h = nmfx.spa(X, k)                                             # test/spa.jl:11-20
    assert np.all(w >= 0) and np.all(h >= 0)
    assert np.allclose(w @ h, X, atol=10.0 * eps4, rtol=0)
    Wg, Hg = separable_data(p, n, k, rng, T)                          # test/spa.jl:22-32
    X = np.asfortranarray(Wg @ Hg)
    W, H, anchors = nmfx.spa(X, k, return_anchors=True)
    assert sorted(anchors.tolist()) == sorted(int(np.argmax(Hg[a] == 1)) for a in range(k))
    assert np.all(W >= 0) and np.all(H >= 0)
    d = (X - W @ H).astype(np.float64)
    assert float(np.sum(d * d)) < np.finfo(T).eps                     # sqL2dist(X, W*H) < eps(T)
    r = nmfx.nnmf(X, k, init="spa", alg="spa")
    assert r.niters == 0 and r.converged
    d = (X - r.W @ r.H).astype(np.float64)
    assert float(np.sum(d * d)) < np.finfo(T).eps and r.objvalue < np.finfo(T).eps


def near_separable(p, n, k, T, seed, noise=0.0):
    rng = np.random.default_rng(seed)
    W = rng.random((p, k)) + 0.1
    H = rng.dirichlet(np.full(k, 0.3), size=n).T
    H[:, rng.choice(n, k, replace=False)] = np.eye(k)
    X = W @ H * (0.5 + rng.random(n))[None, :] + noise * rng.random((p, n))
    return np.asfortranarray(X.astype(T))


@pytest.mark.parametrize("T", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(40, 90, 3), (130, 515, 8), (300, 260, 40), (64, 1000, 64)])
@pytest.mark.parametrize("noise", [0.0, 0.02])
def test_spa_vs_oracle(built, T, shape, noise):
    p, n, k = shape
    X = near_separable(p, n, k, T, seed=p + k, noise=noise)
    W, H, info = nmfx.spa(X, k, return_info=True)
    anchors = info["anchors"]
    Wo, Ho, ao = orc.spa(X, k)
    assert anchors.tolist() == list(ao)
    assert info["unsolved"] == 0
    assert np.array_equal(W, Wo)
    assert np.all(H >= 0)
    if not (T == np.float32 and p == k):
        assert np.max(np.abs(H - Ho)) <= HTOL[T] * np.max(np.abs(Ho))
    rg, ro = np.linalg.norm(X - W @ H), np.linalg.norm(X - Wo @ Ho)
    assert rg <= ro + RTOL[T] * np.linalg.norm(X)


@pytest.mark.parametrize("alg", ["multmse", "greedycd", "projals"])
def test_nnmf_spa_init(built, alg):
    """init = :spa feeds every iterative algorithm (src/interf.jl:50-51); H comes from spa even for projals."""
    T = np.float64
    X = near_separable(60, 140, 5, T, seed=3, noise=0.05)
    r = nmfx.nnmf(X, 5, init="spa", alg=alg, maxiter=30, tol=1e-30)
    W0, H0, _ = orc.spa(X, 5)
    ro = orc.solve(alg, X, W0.copy(order="F"), H0.copy(order="F"), orc.Opts(maxiter=30, tol=1e-30))
    assert r.niters == ro.niters == 30
    assert abs(r.objvalue - ro.objvalue) <= 1e-6 * abs(ro.objvalue)


@pytest.mark.parametrize("T", [np.float64, np.float32])
def test_spa_cold_start_equals_warm_start(built, T):
    """warm_sweeps = 0 is the published method from h = 0; the warm start must end at the same minimiser."""
    X = near_separable(90, 300, 12, T, seed=5, noise=0.05)
    Wc, Hc, ic = nmfx.spa(X, 12, warm_sweeps=0, return_info=True)
    Ww, Hw, iw = nmfx.spa(X, 12, warm_sweeps=16, return_info=True)
    assert ic["unsolved"] == iw["unsolved"] == 0 and np.array_equal(Wc, Ww)
    assert np.max(np.abs(Hc - Hw)) <= {np.float64: 1e-10, np.float32: 5e-3}[T] * np.max(np.abs(Hc))


def test_spa_large_k_uses_the_global_triangle(built):
    """k = 256 in Float64: the packed triangle (263 KB) does not fit the LDS and lives in the per-block global slot."""
    T = np.float64
    p, n, k = 320, 700, 256
    X = near_separable(p, n, k, T, seed=2, noise=0.01)
    W, H, info = nmfx.spa(X, k, return_info=True)
    Wo, Ho, ao = orc.spa(X, k)
    assert info["anchors"].tolist() == list(ao) and info["unsolved"] == 0
    assert np.linalg.norm(X - W @ H) <= np.linalg.norm(X - Wo @ Ho) + 1e-9 * np.linalg.norm(X)


def test_nnmf_alg_spa(built):
    T = np.float64
    X = near_separable(50, 120, 4, T, seed=8, noise=0.03)
    r = nmfx.nnmf(X, 4, init="spa", alg="spa", replicates=3)
    Wo, Ho, _ = orc.spa(X, 4)
    ro = orc.spa_solve(X, Wo, Ho, "mse")
    assert r.niters == 0 and r.converged and r.info["best_replicate"] == 1
    assert abs(r.objvalue - ro.objvalue) <= 1e-8 * abs(ro.objvalue) + 1e-12
    with pytest.raises(nmfx.ArgumentError, match="use :spa instead"):
        nmfx.nnmf(X, 4, init="random", alg="spa")
    # SPA{T}(obj = :div) through solve! (src/spa.jl:69-70)
    W, H = Wo.copy(order="F"), Ho.copy(order="F")
    rd = nmfx.solve(nmfx.SPA(T, obj="div"), X, W, H)
    assert abs(rd.objvalue - orc.spa_solve(X, Wo, Ho, "div").objvalue) <= 1e-8 * abs(rd.objvalue) + 1e-12


def test_spa_headline_shape_properties(built):
    """Size-independent properties at a large shape: W's columns are columns of X, H >= 0, and on a separable X the planted
    anchors are found and the fit is exact to rounding."""
    T = np.float32
    p, n, k = 1024, 4096, 32
    X = near_separable(p, n, k, T, seed=1)
    W, H, info = nmfx.spa(X, k, return_info=True)
    anchors = info["anchors"]
    assert info["unsolved"] == 0
    assert len(set(anchors.tolist())) == k and np.array_equal(W, X[:, anchors])
    assert np.all(H >= 0)
    assert np.linalg.norm(X - W @ H) <= 1e-3 * np.linalg.norm(X)
