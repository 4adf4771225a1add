"""Awkward shapes through every algorithm and both dtypes against the CPU oracle: the smallest possible problem (1 x 1, k = 1),
k = min(p, n), skinny and flat X, sizes that are not multiples of any tile (the device pads to 256 / 128 / 64 internally; padding
must stay inert), k just below / at a tile boundary.  (The larger sweep lives in tests/stress/ and is run by hand.)"""
import numpy as np
import pytest

import c_oracle as co
import nmf_oracle as orc
import nmfx
from problems import rel_trace_err, uniform

pytestmark = pytest.mark.gpu
SHAPES = [(1, 1, 1), (2, 3, 1), (7, 5, 5), (33, 1000, 3), (1000, 33, 17), (129, 257, 128), (129, 257, 100), (300, 263, 65)]


def _inst(alg, T, iters, lam):
    if alg in ("multmse", "multdiv"):
        return nmfx.MultUpdate(T, obj=alg[4:], maxiter=iters, tol=1e-30, lambda_w=lam, lambda_h=lam)
    if alg == "projals":
        return nmfx.ProjectedALS(T, maxiter=iters, tol=1e-30, lambda_w=lam, lambda_h=lam)
    if alg == "cd":
        return nmfx.CoordinateDescent(T, maxiter=iters, tol=1e-30)
    if alg == "greedycd":
        return nmfx.GreedyCD(T, maxiter=iters, tol=1e-30, lambda_w=lam, lambda_h=lam)
    return nmfx.ALSPGrad(T, maxiter=iters, tol=1e-30, maxsubiter=15)


@pytest.mark.parametrize("alg", ["multmse", "multdiv", "projals", "alspgrad", "cd", "greedycd"])
@pytest.mark.parametrize("T", [np.float64, np.float32])
@pytest.mark.parametrize("shape", SHAPES)
def test_edge_shapes(built, alg, T, shape):
    p, n, k = shape
    X, W0, H0 = uniform(p, n, k, T, seed=p + n + k)
    iters = 3 if alg == "alspgrad" else 5
    lam = 0.5 if alg == "projals" else 1e-4
    W, H = W0.copy(order="F"), H0.copy(order="F")
    r = nmfx.solve(_inst(alg, T, iters, lam), X, W, H, track_objective=True)
    Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
    o = orc.Opts(maxiter=iters, tol=1e-30, lambda_w=lam, lambda_h=lam, track_objective=True, maxsubiter=15)
    ro = (co if alg in ("cd", "greedycd") else orc).solve(alg, X, Wc, Hc, o)
    assert r.niters == ro.niters                                  # (the 1 x 1 problem reaches a fixed point and stops early, on both sides)
    loose = alg in ("projals", "alspgrad", "cd", "greedycd")
    tol = {np.float32: (2e-1 if alg == "greedycd" else 5e-4) if loose else 2e-5, np.float64: 1e-7 if loose else 1e-10}[T]
    # an exact fit (the 1 x 1 problem) leaves 0 or one ulp of the factors squared: below the objective's resolution both are "0"
    floor = (8 * np.finfo(T).eps * np.linalg.norm(X.astype(np.float64))) ** 2
    assert rel_trace_err(r.trace, ro.trace, floor) < tol
    assert np.all(W >= 0) and np.all(H >= 0) and np.isfinite(W).all() and np.isfinite(H).all()
