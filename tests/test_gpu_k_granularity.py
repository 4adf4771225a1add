"""Component padding in multiples of 64 (round 4; multiples of 128 above 64 before, so k = 129 ... 192 paid for 256 components in
every product: VERDICT round 3, item 9).  K = 192 / 320 take the general kernel sequence on half-width tiles -- every algorithm
against the CPU oracle at such k, against the old padding (NMFX_K_GRANULE=128: same iterates to rounding), sharded, and through the
exported SPD utilities."""
import numpy as np
import pytest

import nmf_oracle as orc
import nmfx
from problems import planted, rel_trace_err
from test_gpu_localcomm import ALG, lam_for, run_sharded

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("alg", ["multmse", "multdiv", "projals", "alspgrad", "cd", "greedycd"])
@pytest.mark.parametrize("T,k", [(np.float64, 130), (np.float64, 190), (np.float32, 150), (np.float64, 270)])
def test_sixty_four_granular_k_against_the_oracle(built, alg, T, k, monkeypatch):
    p, n = (420, 530) if k < 200 else (640, 700)
    X, W0, H0 = planted(p, n, k, T, seed=31 + k, normalize=(alg != "projals"), k0=min(k, 40))
    # (projals in Float32: a regularisation that keeps the rank-40 Grams well conditioned, as in test_projals_k_beyond_one_lds_column)
    lam = (0.5 if T == np.float64 else 20.0) if alg == "projals" else lam_for(alg, T)
    iters = (2 if k > 200 else 3) if alg == "alspgrad" else 6     # (alspgrad: the CPU oracle's ~600 inner iterations per outer one dominate the test's time)
    kw = dict(maxiter=iters, tol=1e-30, lambda_w=lam, lambda_h=lam, track_objective=True)
    Wg, Hg = W0.copy(order="F"), H0.copy(order="F")
    with nmfx.Context(T, p, n, k) as ctx:
        ctx.set_X(X)
        r, tr = ctx.solve(ALG[alg], nmfx.make_opts(T, **kw), Wg, Hg)
    Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
    ro = orc.solve(alg, X, Wc, Hc, orc.Opts(maxiter=iters, tol=1e-30, lambda_w=lam, lambda_h=lam, track_objective=True))
    assert r.niters == ro.niters == iters
    # (f32 cd: the rank-40 planted problem with k = 150 components leaves entries of H near 2e3; 3e-3 of that measured on the factors)
    tol = {np.float64: 1e-7, np.float32: 2e-3 if alg in ("projals", "alspgrad") else (1e-4 if alg == "cd" else 2e-5)}[T]
    if alg == "greedycd" and T == np.float32:
        # the fp32 greedy sweep is chaotic (two CPU restatements differ by 6e-3 ... 5e-1 after 6 iterations on such
        # problems): descent and the objective's order of magnitude are what can be asserted
        assert np.all(np.diff(tr) <= 1e-3 * tr[:-1]) and tr[-1] < 2.0 * ro.trace[-1] + 1e-6 * tr[0]
    else:
        assert rel_trace_err(tr, ro.trace) < tol
        assert np.max(np.abs(Wg - Wc)) <= 100 * tol * np.max(np.abs(Wc))
        assert np.max(np.abs(Hg - Hc)) <= 100 * tol * np.max(np.abs(Hc))
    assert np.all(Wg >= 0) and np.all(Hg >= 0)
    # the old padding (K = 256 / 384): the same iteration to rounding
    monkeypatch.setenv("NMFX_K_GRANULE", "128")
    Wo, Ho = W0.copy(order="F"), H0.copy(order="F")
    with nmfx.Context(T, p, n, k) as ctx:
        ctx.set_X(X)
        r2, tr2 = ctx.solve(ALG[alg], nmfx.make_opts(T, **kw), Wo, Ho)
    assert r2.niters == iters and (rel_trace_err(tr, tr2) < tol or (alg == "greedycd" and T == np.float32))


@pytest.mark.parametrize("alg", ["multmse", "projals", "alspgrad", "greedycd"])
def test_sixty_four_granular_k_sharded(built, alg):
    T = np.float64
    p, n, k = 300, 530, 150
    X, W0, H0 = planted(p, n, k, T, seed=17, normalize=(alg != "projals"), k0=20)
    lam = 0.5 if alg == "projals" else lam_for(alg, T)
    iters = 3 if alg == "alspgrad" else 5
    kw = dict(maxiter=iters, tol=1e-30, lambda_w=lam, lambda_h=lam, track_objective=True)
    Ws, Hs, rr, Wall = run_sharded(T, X, W0, H0, alg, kw, 2)
    assert np.array_equal(Wall[0], Wall[1])
    Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
    ro = orc.solve(alg, X, Wc, Hc, orc.Opts(maxiter=iters, tol=1e-30, lambda_w=lam, lambda_h=lam, track_objective=True))
    assert rel_trace_err(rr[0][1], ro.trace) < 1e-7
    assert np.max(np.abs(Ws - Wc)) <= 1e-5 * np.max(np.abs(Wc))


@pytest.mark.parametrize("T", [np.float64, np.float32])
def test_pdsolve_identities_at_k_192(built, T):
    """test/utils.jl:48-63 at a k whose padding is 192: x = inv(A) b and x = a inv(B)."""
    k, n, p = 180, 300, 260
    rng = np.random.default_rng(4)
    A = rng.random((k, k)).astype(T)
    A = np.asfortranarray(A @ A.T + k * np.eye(k, dtype=T))
    Bm = np.asfortranarray(rng.random((k, n)).astype(T))
    Am = np.asfortranarray(rng.random((p, k)).astype(T))
    with nmfx.Context(T, p, n, k) as ctx:
        Xl = ctx.pdsolve(A, Bm)
        Xr = ctx.pdrsolve(Am, A)
    tol = 1e-10 if T == np.float64 else 2e-4
    assert np.max(np.abs(A @ Xl - Bm)) <= tol * np.max(np.abs(Bm)) * k
    assert np.max(np.abs(Xr @ A - Am)) <= tol * np.max(np.abs(Am)) * k
