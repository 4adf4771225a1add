"""nnmf front end on the device: non-negativity checks (src/interf.jl:15,28,31), randinit
(src/initialization.jl:4-17) and solve_replicates! (src/interf.jl:85-101)."""
import numpy as np
import pytest

import philox_ref
from problems import uniform


def test_philox_known_answers():
    # Random123 known-answer vectors for philox4x32-10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = philox_ref.philox4x32_10(*[np.array([c]) for c in ctr], *key)
        assert tuple(int(g[0]) for g in got) == want


def test_twin_statistics_and_sharding():
    W, H = philox_ref.randinit(np.float64, 50, 40, 7, seed=5, normalize=True)
    assert W.min() >= 0 and H.min() >= 0 and H.max() < 1
    np.testing.assert_allclose(W.sum(axis=0), 1.0, rtol=1e-14)
    assert abs(H.mean() - 0.5) < 0.05
    # a column shard draws the same numbers as the corresponding slice of the global H
    _, Hs = philox_ref.randinit(np.float64, 50, 15, 7, seed=5, h_col_offset=25)
    np.testing.assert_array_equal(Hs, H[:, 25:40])
    _, Hz = philox_ref.randinit(np.float32, 50, 40, 7, seed=5, zeroh=True)
    assert not Hz.any()


@pytest.mark.gpu
@pytest.mark.parametrize("T", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(37, 53, 5), (300, 129, 64), (257, 700, 130)])
def test_randinit_matches_twin(built, T, shape):
    import nmfx
    p, n, k = shape
    with nmfx.Context(T, p, n, k) as ctx:
        for normalize, zeroh, off in [(False, False, 0), (True, False, 11), (True, True, 0)]:
            ctx.randinit(1234, normalize=normalize, zeroh=zeroh, h_col_offset=off)
            W = np.empty((p, k), dtype=T, order="F"); H = np.empty((k, n), dtype=T, order="F")
            ctx.get_factors(W, H)
            Wr, Hr = philox_ref.randinit(T, p, n, k, 1234, normalize=normalize, zeroh=zeroh, h_col_offset=off)
            np.testing.assert_array_equal(H, Hr)               # bit-exact: same integer stream, same conversion
            if normalize:
                np.testing.assert_allclose(W, Wr, rtol=4 * np.finfo(T).eps)   # column sums: different summation order
                np.testing.assert_allclose(W.sum(axis=0, dtype=np.float64), 1.0, rtol=64 * np.finfo(T).eps)
            else:
                np.testing.assert_array_equal(W, Wr)


@pytest.mark.gpu
@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_check_nonneg(built, T):
    import nmfx
    p, n, k = 130, 70, 9
    X, W0, H0 = uniform(p, n, k, T, seed=3)
    with nmfx.Context(T, p, n, k) as ctx:
        ctx.set_X(X); ctx.set_factors(W0, H0)
        assert ctx.check_nonneg(0) and ctx.check_nonneg(1) and ctx.check_nonneg(2)
        Xn = X.copy(order="F"); Xn[p - 1, n - 1] = -1e-30
        ctx.set_X(Xn)
        assert not ctx.check_nonneg(0)
        Xn[p - 1, n - 1] = np.nan                                # all(t -> t >= 0) is false for NaN too
        ctx.set_X(Xn)
        assert not ctx.check_nonneg(0)
        Xz = X.copy(order="F"); Xz[0, 0] = 0.0; Xz[5, 5] = -0.0   # -0.0 >= 0 holds
        ctx.set_X(Xz)
        assert ctx.check_nonneg(0)
        Hn = H0.copy(order="F"); Hn[k - 1, 0] = -1.0
        ctx.set_factors(W0, Hn)
        assert ctx.check_nonneg(1) and not ctx.check_nonneg(2)


@pytest.mark.gpu
@pytest.mark.parametrize("alg,T", [("multmse", np.float32), ("multdiv", np.float64), ("projals", np.float64), ("alspgrad", np.float64)])
def test_solve_replicates_is_min_over_restarts(built, alg, T):
    """solve_replicates! = replicate 1 from the given start, 2..R from randinit(seed + r - 1); strictly smaller objective wins."""
    import nmfx
    p, n, k, R, seed = 60, 90, 4, 4, 77
    X, W0, H0 = uniform(p, n, k, T, seed=21)
    W0 = np.asfortranarray(W0 / W0.sum(axis=0, keepdims=True))
    zeroh = alg == "projals"
    if alg in ("multmse", "multdiv"):
        inst = nmfx.MultUpdate(T, obj=alg[4:], maxiter=15, tol=1e-12)
    elif alg == "projals":
        inst = nmfx.ProjectedALS(T, maxiter=15, tol=1e-12)
    else:
        inst = nmfx.ALSPGrad(T, maxiter=6, tol=1e-12, maxsubiter=20)
    with nmfx.Context(T, p, n, k) as ctx:
        ctx.set_X(X)
        runs = []
        for r in range(1, R + 1):
            if r == 1:
                W, H = W0.copy(order="F"), H0.copy(order="F")
            else:
                ctx.randinit(seed + r - 1, normalize=True, zeroh=zeroh)
                W = np.empty((p, k), dtype=T, order="F"); H = np.empty((k, n), dtype=T, order="F")
                ctx.get_factors(W, H)
            res = ctx.solve(inst._alg(), nmfx.make_opts(T, **inst._opts()), W, H)[0]
            runs.append((res.objvalue, W, H, res.niters))
        best = 0
        for r in range(1, R):
            if runs[best][0] > runs[r][0]:
                best = r
        W, H = W0.copy(order="F"), H0.copy(order="F")
        res, bi = ctx.solve_replicates(inst._alg(), nmfx.make_opts(T, **inst._opts()), R, seed, zeroh, W, H)
        assert bi == best + 1
        assert res.objvalue == runs[best][0] and res.niters == runs[best][3]
        np.testing.assert_array_equal(W, runs[best][1])
        np.testing.assert_array_equal(H, runs[best][2])
        assert len({round(r[0], 12) for r in runs}) > 1          # the restarts really differ


@pytest.mark.gpu
def test_nnmf_device_front_end(built):
    import nmfx
    T = np.float64
    X, _, _ = uniform(48, 64, 3, T, seed=9)
    r1 = nmfx.nnmf(X, 3, init="random", alg="multmse", maxiter=40, replicates=3, seed=11)
    r2 = nmfx.nnmf(X, 3, init="random", alg="multmse", maxiter=40, replicates=3, seed=11)
    assert r1.objvalue == r2.objvalue and np.array_equal(r1.W, r2.W)            # reproducible from the seed
    assert 1 <= r1.info["best_replicate"] <= 3
    single = nmfx.nnmf(X, 3, init="random", alg="multmse", maxiter=40, replicates=1, seed=11)
    assert r1.objvalue <= single.objvalue
    assert (r1.W >= 0).all() and (r1.H >= 0).all()
    np.testing.assert_allclose(0.5 * np.sum((X - r1.W @ r1.H) ** 2), r1.objvalue, rtol=1e-10)
    Xn = X.copy(); Xn[3, 4] = -0.5
    with pytest.raises(nmfx.ArgumentError, match="elements of X must be non-negative"):
        nmfx.nnmf(Xn, 3, init="random", alg="multmse", seed=1)
    W0 = np.abs(np.random.default_rng(0).random((48, 3))); H0 = np.random.default_rng(1).random((3, 64)); H0[1, 1] = -1
    with pytest.raises(nmfx.ArgumentError, match="elements of H0 must be non-negative"):
        nmfx.nnmf(X, 3, init="custom", alg="multmse", W0=W0, H0=H0, seed=1)
    # init = :custom updates the caller's W0 / H0 in place on both front ends, like the reference (README: "may be overwritten")
    for sd in (None, 4):
        Wc = np.asfortranarray(np.random.default_rng(0).random((48, 3))); Hc = np.asfortranarray(np.random.default_rng(1).random((3, 64)))
        Wb, Hb = Wc.copy(), Hc.copy()
        rc = nmfx.nnmf(X, 3, init="custom", alg="multmse", W0=Wc, H0=Hc, maxiter=10, seed=sd)
        assert rc.W is Wc and rc.H is Hc and not np.array_equal(Wc, Wb) and not np.array_equal(Hc, Hb)
    # projals starts from H = 0 (src/interf.jl:39): the front end must not draw H
    rp = nmfx.nnmf(X, 3, init="random", alg="projals", maxiter=10, seed=5)
    assert np.isfinite(rp.objvalue)


# ---- NNDSVD from a given SVD (src/initialization.jl:26-137; SURVEY.md section 8f rank 3, the part behind `U, s, V = ...`)

def _u01(T, w0, w1):
    if np.dtype(T) == np.float32:
        return np.float32(w0 >> np.uint32(8)) * np.float32(1.0 / 16777216.0)
    return (np.float64(w0 >> np.uint32(5)) * 67108864.0 + np.float64(w1 >> np.uint32(6))) * (1.0 / 9007199254740992.0)


def _rand_vj(T, k, seed):
    """the k uniforms nndsvd_coef_kernel draws for variant :ar (counter = (j, 0, 2, 0))."""
    j = np.arange(k, dtype=np.uint64)
    w0, w1, _, _ = philox_ref.philox4x32_10(j, np.zeros(k, np.uint64), np.full(k, 2, np.uint64), np.zeros(k, np.uint64),
                                            seed & 0xFFFFFFFF, seed >> 32)
    return np.array([_u01(T, a, b) for a, b in zip(w0, w1)], dtype=T)


@pytest.mark.parametrize("T", [np.float64, np.float32])
def test_nndsvd_oracle_kats(T):
    """test/initialization.jl:29-53 on the restatement (with initdata = an SVD of X, as :45-49 do)."""
    import nmf_oracle as orc
    rng = np.random.default_rng(7)
    X = np.asfortranarray(rng.random((8, 12)).astype(T))

    def svd(A):
        U, s, Vt = np.linalg.svd(A.astype(np.float64), full_matrices=False)
        return U.astype(T), s.astype(T), Vt.T.astype(T)

    W, H = orc.nndsvd(X, 5, initdata=svd(X))
    assert W.shape == (8, 5) and H.shape == (5, 12) and (W >= 0).all() and (H >= 0).all()
    W2, H2 = orc.nndsvd(X, 5, zeroh=True, initdata=svd(X))
    assert np.array_equal(W2, W) and not H2.any()
    W2, H2 = orc.nndsvd(T(2) * X, 5, initdata=svd(T(2) * X))
    tol = 200 * np.finfo(T).eps
    np.testing.assert_allclose(W2, np.sqrt(T(2)) * W, rtol=tol, atol=tol)
    np.testing.assert_allclose(H2, np.sqrt(T(2)) * H, rtol=tol, atol=tol)
    Wr, _ = orc.nndsvd(X, 5, variant="ar", initdata=svd(X), rand_vj=rng.random(5).astype(T) + T(0.01))
    assert (Wr > 0).all()
    with pytest.raises(ValueError):
        orc.nndsvd(X, 5, variant="bogus", initdata=svd(X))


@pytest.mark.gpu
@pytest.mark.parametrize("T", [np.float64, np.float32])
@pytest.mark.parametrize("variant", ["std", "a", "ar"])
@pytest.mark.parametrize("shape", [(8, 12, 5), (300, 129, 20), (130, 700, 64)])
def test_nndsvd_matches_oracle(built, T, variant, shape):
    import nmf_oracle as orc
    import nmfx
    p, n, k = shape
    X, _, _ = uniform(p, n, k, T, seed=p)
    F = nmfx.truncated_svd(X, k)
    seed = 4242
    rv = _rand_vj(T, k, seed)
    for zeroh in (False, True):
        W, H = nmfx.nndsvd(X, k, zeroh=zeroh, variant=variant, initdata=F, seed=seed)
        Wo, Ho = orc.nndsvd(X, k, zeroh=zeroh, variant=variant, initdata=F, rand_vj=rv)
        tol = 64 * np.finfo(T).eps       # column norms: the device sums the squares in Float64, the reference left to right in T
        np.testing.assert_allclose(W, Wo, rtol=tol, atol=tol * np.abs(Wo).max())
        np.testing.assert_allclose(H, Ho, rtol=tol, atol=tol * max(np.abs(Ho).max(), 1e-30))
        assert (W >= 0).all() and (H >= 0).all()
        if zeroh:
            assert not H.any()
        if variant == "ar":
            assert (W > 0).all()


@pytest.mark.gpu
def test_nnmf_with_reference_defaults(built):
    """nnmf(X, k) with the reference's default init (:nndsvdar) and algorithm (:greedycd) (src/interf.jl:4-6), and
    init=:nndsvd with initdata like test/interf.jl:20."""
    import nmfx
    T = np.float64
    X, _, _ = uniform(60, 90, 5, T, seed=12)
    r = nmfx.nnmf(X, 5, init="nndsvdar", alg="greedycd", maxiter=50)
    assert np.isfinite(r.objvalue) and (r.W >= 0).all() and (r.H >= 0).all()
    rr = nmfx.nnmf(X, 5, init="random", alg="greedycd", maxiter=50, seed=1)
    assert r.objvalue < 1.2 * rr.objvalue                       # an SVD-based start is at least in the same league as a random one
    F = nmfx.truncated_svd(X, 5)
    for alg in ("multmse", "multdiv", "projals", "alspgrad", "cd", "greedycd"):
        ra = nmfx.nnmf(X, 5, init="nndsvd", alg=alg, maxiter=20, initdata=F)
        assert np.isfinite(ra.objvalue) and (ra.W >= 0).all() and (ra.H >= 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("T", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(300, 260, 8), (129, 515, 20), (700, 600, 70), (1100, 900, 130)])
def test_rsvd_contract(built, T, shape):
    """rsvd(X, k) on the device (src/initialization.jl:83).  Julia's randn stream is not reproducible, so the contract of the
    randomized SVD is pinned instead: orthonormal U and V, non-negative descending s, and on a matrix of numerical rank k
    (planted + 1 % noise) a reconstruction error close to the optimal rank-k truncation."""
    import nmfx
    from problems import planted
    p, n, k = shape
    X, _, _ = planted(p, n, k, T, seed=3 * p + n)
    X64 = X.astype(np.float64)
    sv = np.linalg.svd(X64, compute_uv=False)
    best = np.sqrt(np.sum(sv[k:] ** 2))
    eps = np.finfo(T).eps
    for q, slack in ((0, None), (1, 1.05)):
        U, s, V = nmfx.rsvd(X, k, seed=99, power_iters=q)
        assert U.shape == (p, k) and V.shape == (n, k) and s.shape == (k,)
        assert np.all(np.diff(s) <= 0) and s[-1] >= 0
        assert np.abs(U.T.astype(np.float64) @ U - np.eye(k)).max() < 200 * eps
        assert np.abs(V.T.astype(np.float64) @ V - np.eye(k)).max() < (2e-3 if T == np.float32 else 1e-9)
        err = np.linalg.norm(X64 - (U.astype(np.float64) * s) @ V.T.astype(np.float64))
        if slack is None:
            # plain k-column sketch: the error is a random multiple of the optimum (a NumPy restatement of the same
            # algorithm -- QR of X*Omega, no oversampling -- gives 3x .. 30x on these matrices); bound it by the energy
            assert best <= err * (1 + 1e-9) and err <= 0.1 * sv[0]
        else:
            # one power iteration on a matrix with a spectral gap recovers the optimal rank-k error
            assert err <= slack * best + 50 * eps * sv[0]
            np.testing.assert_allclose(s, sv[:k], rtol=1e-3 if T == np.float32 else 1e-6)
    # same seed -> same sketch: the device half is bit-reproducible, the host LAPACK step in the middle need not be (its
    # eigenvectors of the nearly degenerate noise-level eigenvalues move with the last bit), so compare what is well defined
    U2, s2, V2 = nmfx.rsvd(X, k, seed=99, power_iters=1)
    np.testing.assert_allclose(s2, s, rtol=1e-4 if T == np.float32 else 1e-9)
    A1 = (U.astype(np.float64) * s) @ V.T.astype(np.float64)
    A2 = (U2.astype(np.float64) * s2) @ V2.T.astype(np.float64)
    assert np.linalg.norm(A1 - A2) <= (1e-4 if T == np.float32 else 1e-9) * sv[0]


@pytest.mark.gpu
def test_nndsvd_default_svd_is_the_device_rsvd(built):
    import nmfx
    T = np.float64
    X, _, _ = uniform(90, 120, 6, T, seed=4)
    W, H = nmfx.nndsvd(X, 6)                      # no initdata: rsvd on the device, result never leaves it
    assert W.shape == (90, 6) and H.shape == (6, 120) and (W >= 0).all() and (H >= 0).all()
    W2, H2 = nmfx.nndsvd(X, 6, zeroh=True)
    assert np.array_equal(W2, W) and not H2.any()  # test/initialization.jl:37-43 (same seed)
    Wr, _ = nmfx.nndsvd(X, 6, variant="ar")
    assert (Wr > 0).all()
    # leading component: close to the exact SVD's (the randomized sketch captures the dominant direction)
    We, _ = nmfx.nndsvd(X, 6, initdata=nmfx.truncated_svd(X, 6))
    Wp, _ = nmfx.nndsvd(X, 6, power_iters=3)
    assert np.abs(Wp[:, 0] - We[:, 0]).max() < 0.02 * np.abs(We[:, 0]).max()


@pytest.mark.gpu
def test_nnmf_reference_defaults(built):
    """nnmf(X, k) with NO keywords = the reference's defaults init = :nndsvdar, alg = :greedycd (src/interf.jl:4-7); the
    non-negativity check of X (src/interf.jl:15) then runs on the device."""
    import nmfx
    rng = np.random.default_rng(12)
    X = np.asfortranarray(rng.random((60, 90)))
    r = nmfx.nnmf(X, 4)
    assert r.niters >= 1 and np.isfinite(r.objvalue) and np.all(r.W >= 0) and np.all(r.H >= 0)
    r2 = nmfx.nnmf(X, 4, init="nndsvdar", alg="greedycd")
    assert r == r2
    Xneg = X.copy()
    Xneg[3, 4] = -0.5
    with pytest.raises(nmfx.ArgumentError, match="non-negative"):
        nmfx.nnmf(Xneg, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("T", [np.float64, np.float32])
@pytest.mark.parametrize("variant", ["std", "a"])
@pytest.mark.parametrize("shape", [(40, 60, 5), (300, 129, 20), (130, 700, 64)])
def test_nndsvd_is_bit_identical_on_dyadic_singular_vectors(built, T, variant, shape):
    """_nndsvd! (src/initialization.jl:26-72) from "singular vectors" with dyadic entries: the sums of squares of posnegnorm
    (:103-115) are exact in T in any order (the device sums them in Float64, the reference left to right in T), and what follows
    -- two square roots, the products s_j * mp, the quotients ss / xp, one product per element, the mean of an integer X for
    variant :a -- is correctly rounded element-wise arithmetic: W and H must equal the oracle's BIT FOR BIT."""
    import nmf_oracle as orc
    import nmfx
    p, n, k = shape
    rng = np.random.default_rng(p + k)
    X = np.asfortranarray(rng.integers(0, 8, size=(p, n)).astype(T))
    U = np.asfortranarray((rng.integers(-6, 7, size=(p, k)) / 8.0).astype(T))
    V = np.asfortranarray((rng.integers(-6, 7, size=(n, k)) / 8.0).astype(T))
    s = np.sort(rng.integers(1, 40, size=k)).astype(T)[::-1].copy()
    for zeroh in (False, True):
        W, H = nmfx.nndsvd(X, k, zeroh=zeroh, variant=variant, initdata=(U, s, V))
        Wo, Ho = orc.nndsvd(X, k, zeroh=zeroh, variant=variant, initdata=(U, s, V))
        Ui = np.uint32 if T == np.float32 else np.uint64
        assert np.array_equal(W.view(Ui), Wo.view(Ui)), float(np.max(np.abs(W - Wo)))
        assert np.array_equal(H.view(Ui), Ho.view(Ui)), float(np.max(np.abs(H - Ho)))
