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
    assert abs(r.niters - ro.niters) <= 1
    assert len(r.trace) == r.niters + 1 and np.all(np.isfinite(r.trace))
    assert r.objvalue == r.trace[-1]
    m = min(len(r.trace), len(ro.trace))
    np.testing.assert_allclose(r.trace[:m], np.array(ro.trace)[:m], rtol=1e-6)
    # and without tracking the same final value comes from the single evaluation at the end (common.jl:85-87)
    W2, H2 = W0.copy(order="F"), H0.copy(order="F")
    r2 = nmfx.solve(_inst(alg, T, maxiter=400, tol=tol), X, W2, H2)
    assert r2.niters == r.niters and r2.objvalue == r.objvalue and np.array_equal(W2, W)
