"""The two CPU restatements of the reference (oracle/nmf_oracle.py: NumPy / LAPACK; oracle/nmf_oracle.c: plain C loops) against each
other BIT FOR BIT on inputs whose products and sums are exact in T: there the GEMMs' summation order (OpenBLAS kernels vs simple
loops) cannot matter, and what is left is the element-wise arithmetic of the update rules -- which both restate from the same
lines of the reference (multupd.jl:98-115, 172-191; coorddesc.jl:107-158; greedycd.jl:91-163; alspgrad.jl:86-191).  The GPU
tests of the same name pattern (`*exact*`, `*bit_identical*`) compare the device with these oracles on the same constructions."""
import numpy as np
import pytest

import c_oracle as co
import nmf_oracle as orc


def _ints(T, p, n, k, seed, hi=(4, 3, 3)):
    rng = np.random.default_rng(seed)
    return (np.asfortranarray(rng.integers(0, hi[0], size=(p, n)).astype(T)), np.asfortranarray(rng.integers(0, hi[1], size=(p, k)).astype(T)),
            np.asfortranarray(rng.integers(0, hi[2], size=(k, n)).astype(T)), rng)


def _same(a, b):
    return np.array_equal(a.view(np.uint32 if a.dtype == np.float32 else np.uint64), b.view(np.uint32 if b.dtype == np.float32 else np.uint64))


@pytest.mark.parametrize("T", [np.float32, np.float64])
@pytest.mark.parametrize("update_H", [True, False])
def test_multmse_first_update(T, update_H):
    X, W0, H0, _ = _ints(T, 130, 170, 40, 1)
    o = orc.Opts(maxiter=1, tol=1e-30, update_H=update_H)
    Wa, Ha, Wb, Hb = W0.copy(order="F"), H0.copy(order="F"), W0.copy(order="F"), H0.copy(order="F")
    orc.solve("multmse", X, Wa, Ha, o)
    co.solve("multmse", X, Wb, Hb, o)
    assert _same(Ha, Hb) if update_H else (_same(Wa, Wb) and np.array_equal(Ha, H0))
    assert not np.array_equal(Ha if update_H else Wa, H0 if update_H else W0)


@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_multdiv_first_h_update_on_selection_factors(T):
    p, n, k = 200, 90, 30
    X, _, H0, rng = _ints(T, p, n, k, 2, hi=(50, 3, 7))
    W0 = np.zeros((p, k), dtype=T, order="F")
    W0[rng.permutation(p)[:k], np.arange(k)] = 1
    o = orc.Opts(maxiter=1, tol=1e-30, lambda_w=0.25, lambda_h=0.25)
    Wa, Ha, Wb, Hb = W0.copy(order="F"), H0.copy(order="F"), W0.copy(order="F"), H0.copy(order="F")
    orc.solve("multdiv", X, Wa, Ha, o)
    co.solve("multdiv", X, Wb, Hb, o)
    assert _same(Ha, Hb) and not np.array_equal(Ha, H0)


@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_greedycd_first_w_sweep(T):
    X, W0, H0, _ = _ints(T, 100, 120, 70, 3)
    o = orc.Opts(maxiter=1, tol=1e-30, update_H=False, lambda_w=0.25, lambda_h=0.25)
    Wa, Ha, Wb, Hb = W0.copy(order="F"), H0.copy(order="F"), W0.copy(order="F"), H0.copy(order="F")
    ra = orc.solve("greedycd", X, Wa, Ha, o)
    rb = co.solve("greedycd", X, Wb, Hb, o)
    assert _same(Wa, Wb) and ra.counters["greedy_steps"] == rb.counters["inner"] > 0


@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_cd_first_w_sweep_with_a_diagonal_gram(T):
    p, n, k = 150, 90, 40
    X, _, _, rng = _ints(T, p, n, k, 4, hi=(9, 3, 3))
    W0 = np.asfortranarray((rng.integers(0, 40, size=(p, k)) / 8.0).astype(T))
    H0 = np.zeros((k, n), dtype=T, order="F")
    H0[np.arange(k), rng.permutation(n)[:k]] = 1
    o = orc.Opts(maxiter=1, tol=1e-30, update_H=False, l1_w=0.25, l2_w=0.5, l1_h=0.25, l2_h=0.5)
    Wa, Ha, Wb, Hb = W0.copy(order="F"), H0.copy(order="F"), W0.copy(order="F"), H0.copy(order="F")
    orc.solve("cd", X, Wa, Ha, o)
    co.solve("cd", X, Wb, Hb, o)
    assert _same(Wa, Wb) and not np.array_equal(Wa, W0)


@pytest.mark.parametrize("T", [np.float32, np.float64])
@pytest.mark.parametrize("side", ["h", "w"])
def test_alspgrad_first_projected_gradient_step(T, side):
    X, W0, H0, _ = _ints(T, 110, 140, 50, 5)
    Wa, Ha, Wb, Hb = W0.copy(order="F"), H0.copy(order="F"), W0.copy(order="F"), H0.copy(order="F")
    f = "alspgrad_updateh" if side == "h" else "alspgrad_updatew"
    na = getattr(orc, f)(X, Wa, Ha, maxiter=1, tolg=1e-30)
    nb = getattr(co, f)(X, Wb, Hb, maxiter=1, tolg=1e-30)
    assert na == nb == 1
    assert _same(Ha, Hb) and _same(Wa, Wb)
    assert not np.array_equal(Ha if side == "h" else Wa, H0 if side == "h" else W0)
