"""test/utils.jl:6-15, 29-34, 48-63 run through the C ABI on the device kernels ProjectedALS is built from
(nmfx_pdsolve / nmfx_pdrsolve: adddiag! + blocked Cholesky + triangular inverse + MFMA products [+ projectnn!]) --
SURVEY.md section 8 row a16."""
import numpy as np
import pytest

import nmfx

pytestmark = pytest.mark.gpu


def make_pdmat(n, rng, T):
    g = rng.standard_normal((n, n))
    return np.asfortranarray((g.T @ g + 0.1 * np.eye(n)).astype(T))       # make_pdmat of test/utils.jl:4


@pytest.mark.parametrize("T", [np.float64, np.float32])
# (k = 300, 500: the Float32 strip kernels with 10 and 16 block rows; Float64 takes the panel kernel / the product form there)
@pytest.mark.parametrize("k,cols,rows", [(5, 3, 4), (32, 40, 33), (70, 130, 129), (256, 300, 260), (300, 700, 310), (500, 900, 520)])
def test_pdsolve_pdrsolve_identities(built, T, k, cols, rows):
    rng = np.random.default_rng(k)
    tol = 1e-9 if T == np.float64 else 2e-3
    with nmfx.Context(T, rows, cols, k) as ctx:
        A = make_pdmat(k, rng, T)                                          # pdsolve!: X == inv(A) (A X)   (test/utils.jl:48-52)
        X = rng.random((k, cols)).astype(T)
        Y = np.asfortranarray(A @ X)
        Xs = ctx.pdsolve(A, Y)
        assert np.max(np.abs(Xs - X)) <= tol * max(1.0, np.linalg.cond(A.astype(np.float64)) * 1e-2)
        B = make_pdmat(k, rng, T)                                          # pdrsolve!: Xr == (X B) inv(B)  (test/utils.jl:56-61)
        Xw = rng.random((rows, k)).astype(T)
        Yw = np.asfortranarray(Xw @ B)
        Xr = ctx.pdrsolve(Yw, B)
        assert np.max(np.abs(Xr - Xw)) <= tol * max(1.0, np.linalg.cond(B.astype(np.float64)) * 1e-2)


@pytest.mark.parametrize("T", [np.float64, np.float32])
def test_adddiag_and_projectnn_semantics(built, T):
    """adddiag!(A, 0) is a no-op, adddiag!(A, a) adds a*I (test/utils.jl:8-15); projectnn! clamps negatives, keeps the rest
    (test/utils.jl:29-34) -- observed through the solves they feed in projals."""
    rng = np.random.default_rng(3)
    k, cols, rows = 6, 9, 7
    with nmfx.Context(T, rows, cols, k) as ctx:
        A = make_pdmat(k, rng, T)
        B = np.asfortranarray(rng.standard_normal((k, cols)).astype(T))
        ref0 = np.linalg.solve(A.astype(np.float64), B.astype(np.float64))
        ref1 = np.linalg.solve(A.astype(np.float64) + 2.5 * np.eye(k), B.astype(np.float64))
        tol = 1e-10 if T == np.float64 else 1e-3
        assert np.allclose(ctx.pdsolve(A, B, 0.0), ref0, atol=tol * np.abs(ref0).max())
        assert np.allclose(ctx.pdsolve(A, B, 2.5), ref1, atol=tol * np.abs(ref1).max())
        Xp = ctx.pdsolve(A, B, 2.5, project_nn=True)
        assert (ref1 < 0).any() and np.all(Xp >= 0)
        assert np.allclose(Xp, np.maximum(ref1, 0.0), atol=tol * np.abs(ref1).max())
        Aw = np.asfortranarray(rng.standard_normal((rows, k)).astype(T))
        refw = Aw.astype(np.float64) @ np.linalg.inv(A.astype(np.float64) + 0.5 * np.eye(k))
        Xw = ctx.pdrsolve(Aw, A, 0.5, project_nn=True)
        assert np.allclose(Xw, np.maximum(refw, 0.0), atol=tol * np.abs(refw).max())


def test_pdsolve_not_posdef(built):
    """potrf! on a matrix that is not positive definite -> PosDefException (src/utils.jl:68)."""
    with nmfx.Context(np.float64, 4, 5, 3) as ctx:
        with pytest.raises(nmfx.PosDefException):
            ctx.pdsolve(-np.eye(3), np.ones((3, 5)))
        # the context stays usable
        X = ctx.pdsolve(2.0 * np.eye(3), np.ones((3, 5)))
        assert np.allclose(X, 0.5)
