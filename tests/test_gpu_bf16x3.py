"""The opt-in mixed-precision form of the two p*n*k products (nmfx_opts.precision = NMFX_PREC_BF16X3, csrc/gemm_bf16x3.hpp;
SURVEY.md section 8f rank 4): operands split into bf16 pairs, hi*hi + hi*lo + lo*hi on the bf16 matrix cores, fp32
accumulation.  It must stay within the SAME tolerances as the fp32 path: objective trajectory within 1e-5 relative of the
CPU oracle for the multiplicative updates (north_star), and the per-algorithm tolerances elsewhere."""
import numpy as np
import pytest

import c_oracle as co
import nmf_oracle as orc
import nmfx
from problems import planted, rel_trace_err, uniform

pytestmark = pytest.mark.gpu
T = np.float32


@pytest.mark.parametrize("obj", ["mse", "div"])
@pytest.mark.parametrize("shape", [(300, 260, 70), (1024, 768, 128), (700, 1500, 200), (2048, 2048, 256)])
def test_multupd_trajectory_within_1e5(built, obj, shape):
    p, n, k = shape
    X, W0, H0 = planted(p, n, k, T, seed=p + k)
    alg = nmfx.MultUpdate(T, obj=obj, maxiter=8, tol=1e-30)
    W, H = W0.copy(order="F"), H0.copy(order="F")
    r = nmfx.solve(alg, X, W, H, track_objective=True, precision="bf16x3")
    Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
    ro = orc.solve("mult" + obj, X, Wc, Hc, orc.Opts(maxiter=8, tol=1e-30, track_objective=True))
    assert r.niters == ro.niters == 8
    assert rel_trace_err(r.trace, ro.trace) < 1e-5
    assert np.max(np.abs(W - Wc)) <= 1e-3 * np.max(np.abs(Wc)) and np.max(np.abs(H - Hc)) <= 1e-3 * np.max(np.abs(Hc))
    assert np.all(W >= 0) and np.all(H >= 0)
    # and it really is a different arithmetic: the fp32 path gives (slightly) different factors
    W2, H2 = W0.copy(order="F"), H0.copy(order="F")
    nmfx.solve(alg, X, W2, H2)
    assert not np.array_equal(W2, W)


@pytest.mark.parametrize("algname", ["projals", "alspgrad", "cd", "greedycd"])
def test_other_algorithms_keep_their_tolerances(built, algname):
    p, n, k = 520, 640, 96
    X, W0, H0 = uniform(p, n, k, T, seed=31)
    if algname == "projals":
        W0 = np.asfortranarray(np.random.default_rng(2).random((p, k)).astype(T))
    mk = {"projals": lambda: nmfx.ProjectedALS(T, maxiter=5, tol=1e-30, lambda_w=0.5, lambda_h=0.5),
          "alspgrad": lambda: nmfx.ALSPGrad(T, maxiter=3, tol=1e-30, maxsubiter=10),
          "cd": lambda: nmfx.CoordinateDescent(T, maxiter=5, tol=1e-30),
          "greedycd": lambda: nmfx.GreedyCD(T, maxiter=4, tol=1e-30)}[algname]
    outs = {}
    for prec in ("fp32", "bf16x3"):
        W, H = W0.copy(order="F"), H0.copy(order="F")
        outs[prec] = nmfx.solve(mk(), X, W, H, track_objective=True, precision=prec)
    kw = dict(lambda_w=0.5, lambda_h=0.5) if algname == "projals" else (dict(maxsubiter=10) if algname == "alspgrad" else {})
    it = {"projals": 5, "alspgrad": 3, "cd": 5, "greedycd": 4}[algname]
    ro = co.solve(algname, X, W0.copy(order="F"), H0.copy(order="F"), orc.Opts(maxiter=it, tol=1e-30, track_objective=True, **kw))
    tol = {"projals": 2e-3, "alspgrad": 2e-3, "cd": 5e-4, "greedycd": 2e-2}[algname]
    e32, e16 = rel_trace_err(outs["fp32"].trace, ro.trace), rel_trace_err(outs["bf16x3"].trace, ro.trace)
    assert e16 < max(tol, 3 * e32), (e32, e16)


def test_small_k_falls_back_to_fp32(built):
    """k <= 64 (K = 64): the option is ignored, results are bit-identical to the fp32 path."""
    X, W0, H0 = planted(300, 280, 20, T, seed=5)
    alg = nmfx.MultUpdate(T, maxiter=5, tol=1e-30)
    W1, H1 = W0.copy(order="F"), H0.copy(order="F")
    W2, H2 = W0.copy(order="F"), H0.copy(order="F")
    nmfx.solve(alg, X, W1, H1, precision="fp32")
    nmfx.solve(alg, X, W2, H2, precision="bf16x3")
    assert np.array_equal(W1, W2) and np.array_equal(H1, H2)


def test_f64_ignores_the_option(built):
    X, W0, H0 = planted(300, 280, 70, np.float64, seed=5)
    alg = nmfx.MultUpdate(np.float64, maxiter=4, tol=1e-30)
    W1, H1 = W0.copy(order="F"), H0.copy(order="F")
    W2, H2 = W0.copy(order="F"), H0.copy(order="F")
    nmfx.solve(alg, X, W1, H1)
    nmfx.solve(alg, X, W2, H2, precision="bf16x3")
    assert np.array_equal(W1, W2) and np.array_equal(H1, H2)
