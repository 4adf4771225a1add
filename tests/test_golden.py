"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py): both CPU restatements must
reproduce them on any machine (different BLAS kernels => tolerance, not bit equality), and on the GPU box
the HIP path must match them through the C ABI."""
import glob
import os

import numpy as np
import pytest

import c_oracle as co
import nmf_oracle as orc
from problems import rel_trace_err

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))
# objective-trajectory tolerance (relative, every iteration): (f64, f32)
TOL_CPU = {"multmse": (1e-11, 5e-6), "multdiv": (1e-11, 5e-6), "projals": (1e-8, 2e-3), "alspgrad": (1e-8, 2e-3),
           "cd": (1e-11, 2e-4), "greedycd": (1e-10, 5e-4)}
TOL_GPU = {"multmse": (1e-10, 1e-5), "multdiv": (1e-10, 1e-5), "projals": (1e-7, 2e-3), "alspgrad": (1e-7, 5e-4),
           "cd": (1e-10, 5e-4), "greedycd": (1e-9, 1e-3)}
ALGS = ("multmse", "multdiv", "projals", "alspgrad", "cd", "greedycd")


def _load(path):
    z = np.load(path)
    alg = str(z["alg"])
    maxiter, tol, lw, lh, delta, tolg = z["opts"]
    kw = dict(maxiter=int(maxiter), tol=float(tol), lambda_w=float(lw), lambda_h=float(lh), delta=float(delta), tolg=float(tolg))
    if "opts_cd" in z.files:
        kw.update(dict(zip(("l1_w", "l2_w", "l1_h", "l2_h"), (float(v) for v in z["opts_cd"]))))
    return z, alg, kw


def test_fixture_inventory():
    names = {os.path.basename(f) for f in FILES}
    assert names == {f"{a}_{d}.npz" for a in ALGS for d in ("float32", "float64")}


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
@pytest.mark.parametrize("impl", ["numpy", "c"])
def test_oracles_reproduce_goldens(path, impl):
    z, alg, kw = _load(path)
    T = z["X"].dtype.type
    W, H = np.asfortranarray(z["W0"].copy()), np.asfortranarray(z["H0"].copy())
    r = (orc if impl == "numpy" else co).solve(alg, np.asfortranarray(z["X"]), W, H, orc.Opts(track_objective=True, **kw))
    tol = TOL_CPU[alg][0 if T == np.float64 else 1]
    assert r.niters == int(z["niters"]) and r.converged == bool(z["converged"])
    assert rel_trace_err(r.trace, z["trace"]) < tol
    assert np.max(np.abs(W - z["W"])) <= 100 * tol * np.max(np.abs(z["W"]))
    assert np.max(np.abs(H - z["H"])) <= 100 * tol * np.max(np.abs(z["H"]))


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_hip_path_reproduces_goldens(built, path):
    import nmfx
    z, alg, kw = _load(path)
    T = z["X"].dtype.type
    X = np.asfortranarray(z["X"])
    W, H = np.asfortranarray(z["W0"].copy()), np.asfortranarray(z["H0"].copy())
    if alg in ("multmse", "multdiv"):
        inst = nmfx.MultUpdate(T, obj=alg[4:], maxiter=kw["maxiter"], tol=kw["tol"], lambda_w=kw["lambda_w"], lambda_h=kw["lambda_h"])
    elif alg == "projals":
        inst = nmfx.ProjectedALS(T, maxiter=kw["maxiter"], tol=kw["tol"], lambda_w=kw["lambda_w"], lambda_h=kw["lambda_h"])
    elif alg == "alspgrad":
        inst = nmfx.ALSPGrad(T, maxiter=kw["maxiter"], tol=kw["tol"], tolg=kw["tolg"])
    elif alg == "cd":
        inst = nmfx.CoordinateDescent(T, maxiter=kw["maxiter"], tol=kw["tol"])
        inst.l1_w, inst.l2_w, inst.l1_h, inst.l2_h = kw["l1_w"], kw["l2_w"], kw["l1_h"], kw["l2_h"]
    else:
        inst = nmfx.GreedyCD(T, maxiter=kw["maxiter"], tol=kw["tol"], lambda_w=kw["lambda_w"], lambda_h=kw["lambda_h"])
    r = nmfx.solve(inst, X, W, H, track_objective=True)
    tol = TOL_GPU[alg][0 if T == np.float64 else 1]
    assert r.niters == int(z["niters"]) and r.converged == bool(z["converged"])
    assert rel_trace_err(r.trace, z["trace"]) < tol
    assert np.max(np.abs(W - z["W"])) <= 200 * tol * np.max(np.abs(z["W"]))
    assert np.max(np.abs(H - z["H"])) <= 200 * tol * np.max(np.abs(z["H"]))
