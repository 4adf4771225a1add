"""Host-side mirror of the reference interface: validation and error vocabulary (no GPU needed:
every check below fires before a device context is created).  src/interf.jl:15-33,55,79; src/multupd.jl:27-40;
src/common.jl:5-16,29-31; test/utils.jl:65-69 (Result ==/hash)."""
import numpy as np
import pytest

import nmfx


def test_nnmf_argument_errors():
    X = np.asfortranarray(np.random.default_rng(0).random((6, 8)))
    Xneg = X.copy()
    Xneg[0, 0] = -1
    with pytest.raises(nmfx.ArgumentError, match="non-negative"):
        nmfx.nnmf(Xneg, 2, init="random")       # (the default init = :nndsvdar checks X on the device: tests/test_frontend.py)
    with pytest.raises(nmfx.ArgumentError, match="should not exceed"):
        nmfx.nnmf(X, 7, init="random")
    with pytest.raises(nmfx.ArgumentError, match="replicates"):
        nmfx.nnmf(X, 2, init="random", replicates=0)
    with pytest.raises(nmfx.ArgumentError, match="set W0 and H0"):
        nmfx.nnmf(X, 2, init="custom")
    W0 = np.ones((6, 2), order="F")
    H0 = np.ones((2, 8), order="F")
    with pytest.raises(nmfx.ArgumentError, match="Invalid size for W0"):
        nmfx.nnmf(X, 2, init="custom", W0=np.ones((5, 2)), H0=H0)
    with pytest.raises(nmfx.ArgumentError, match="Invalid size for H0"):
        nmfx.nnmf(X, 2, init="custom", W0=W0, H0=np.ones((2, 7)))
    with pytest.raises(nmfx.ArgumentError, match="W0 must be non-negative"):
        nmfx.nnmf(X, 2, init="custom", W0=-W0, H0=H0)
    with pytest.raises(nmfx.ArgumentError, match="Invalid value for init"):
        nmfx.nnmf(X, 2, init="bogus")
    with pytest.raises(nmfx.ArgumentError, match="Invalid algorithm"):
        nmfx.nnmf(X, 2, alg="bogus")
    with pytest.raises(nmfx.ArgumentError, match="Invalid value for init, use :spa instead"):      # src/interf.jl:74-76
        nmfx.nnmf(X, 2, init="random", alg="spa")
    with pytest.raises(nmfx.ArgumentError, match="Invalid value for obj"):
        nmfx.SPA(np.float64, obj="bogus")


def test_coordinate_descent_option_structs():
    """CoordinateDescentUpd's l1/l2 resolution (src/coorddesc.jl:62-82) and GreedyCD's validation (src/greedycd.jl:24-29)."""
    c = nmfx.CoordinateDescent(np.float64, alpha=1e-2, l1ratio=0.25, regularization="both")
    assert (c.l1_w, c.l2_w, c.l1_h, c.l2_h) == (0.0025, 0.0075, 0.0025, 0.0075) and abs(c.tol - 6.06e-6) < 1e-8
    c = nmfx.CoordinateDescent(np.float64, alpha=1e-2, l1ratio=0.25, regularization="components")
    assert (c.l1_w, c.l2_w) == (0.0, 0.0) and (c.l1_h, c.l2_h) == (0.0025, 0.0075)
    c = nmfx.CoordinateDescent(np.float64, alpha=1e-2, l1ratio=0.25, regularization="transformation")
    assert (c.l1_w, c.l2_w) == (0.0025, 0.0075) and (c.l1_h, c.l2_h) == (0.0, 0.0)
    c = nmfx.CoordinateDescent(np.float32)
    assert (c.l1_w, c.l2_w, c.l1_h, c.l2_h) == (0.0, 0.0, 0.0, 0.0) and c.maxiter == 100
    with pytest.raises(nmfx.ArgumentError, match="shuffle_seed"):
        nmfx.CoordinateDescent(np.float32, shuffle=True, shuffle_seed=0)
    assert nmfx.CoordinateDescent(np.float32, shuffle=True, shuffle_seed=9)._opts()["cd_shuffle"] == 9
    assert nmfx.CoordinateDescent(np.float32)._opts()["cd_shuffle"] == 0
    g = nmfx.GreedyCD(np.float32)
    assert g.lambda_w == 0 and g.lambda_h == 0 and abs(g.tol - 4.92e-3) < 1e-5
    for bad in (dict(maxiter=1), dict(tol=0), dict(lambda_w=-1), dict(lambda_h=-1)):
        with pytest.raises(nmfx.ArgumentError):
            nmfx.GreedyCD(np.float32, **bad)


def test_option_struct_defaults_and_validation():
    m = nmfx.MultUpdate(np.float32)
    assert m.maxiter == 100 and abs(m.tol - 4.92e-3) < 1e-5 and m.lambda_w == 0 and m.update_H
    d = nmfx.MultUpdate(np.float64, obj="div")
    assert abs(d.lambda_w - 1.4901161193847656e-08) < 1e-20 and d.lambda_h == d.lambda_w       # max(lambda, sqrt(eps))
    assert abs(d._opts()["delta"] - 1.4901161193847656e-08) < 1e-20
    for bad in (dict(obj="x"), dict(maxiter=1), dict(tol=0), dict(lambda_w=-1), dict(lambda_h=-1)):
        with pytest.raises(nmfx.ArgumentError):
            nmfx.MultUpdate(np.float32, **bad)
    with pytest.warns(UserWarning, match="deprecated"):
        dep = nmfx.MultUpdate(np.float64, lambda_=0.5)
    assert dep.lambda_w == 0.5 and dep.lambda_h == 0.5
    pa = nmfx.ProjectedALS(np.float32)
    assert abs(pa.lambda_w - 4.92e-3) < 1e-5 and abs(pa.tol - 4.92e-3) < 1e-5
    ag = nmfx.ALSPGrad(np.float64)
    assert ag.maxsubiter == 200 and abs(ag.tolg - 1.2207e-4) < 1e-8


def test_result_struct():
    W = np.ones((4, 2), order="F")
    H = np.ones((2, 5), order="F")
    a = nmfx.Result(W, H, 3, True, 0.5)
    b = nmfx.Result(W.copy(), H.copy(), 3, True, 0.5)
    assert a == b and hash(a) == hash(b)
    assert not (a == nmfx.Result(W, H, 4, True, 0.5))
    with pytest.raises(nmfx.DimensionMismatch, match="Inner dimensions"):
        nmfx.Result(W, np.ones((3, 5)), 1, False, 0.0)


def test_checksize():
    X = np.zeros((5, 7))
    assert nmfx.nmf_checksize(X, np.zeros((5, 2)), np.zeros((2, 7))) == (5, 7, 2)
    with pytest.raises(nmfx.DimensionMismatch, match="inconsistent"):
        nmfx.nmf_checksize(X, np.zeros((5, 2)), np.zeros((2, 6)))


def test_randinit_properties():
    X = np.zeros((9, 11), dtype=np.float32)
    W, H = nmfx.randinit(X, 3, normalize=True, rng=np.random.default_rng(0))
    assert W.dtype == np.float32 and W.flags.f_contiguous and np.allclose(W.sum(axis=0), 1, atol=1e-6)
    assert H.shape == (3, 11) and (H >= 0).all() and (H < 1).all()
    _, Hz = nmfx.randinit(X, 3, zeroh=True)
    assert not Hz.any()


def test_greedy_division_identity():
    """csrc/cd.hpp::greedy_div replaces the Float32 division on GreedyCD's step chain by (float)((double)g * (1.0 / den)): that IS
    the correctly rounded quotient (argument in the source).  Checked here on wide-range random operands and on operands built to
    put the quotient next to a rounding boundary of the float grid."""
    rng = np.random.default_rng(0)
    n = 2_000_000
    g = (rng.standard_normal(n) * np.exp(rng.uniform(-40, 40, n))).astype(np.float32)
    den = np.exp(rng.uniform(-30, 30, n)).astype(np.float32)
    with np.errstate(over="ignore", under="ignore"):
        q1 = g / den
        q2 = (g.astype(np.float64) * (1.0 / den.astype(np.float64))).astype(np.float32)
    assert np.array_equal(q1.view(np.uint32), q2.view(np.uint32))
    den = (1 + rng.random(n)).astype(np.float32)
    qf = (1 + rng.random(n)).astype(np.float32)
    mid = qf.astype(np.float64) + np.spacing(qf).astype(np.float64) / 2
    g = (mid * den.astype(np.float64)).astype(np.float32)          # the float nearest to (midpoint of two floats) * den
    q1 = g / den
    q2 = (g.astype(np.float64) * (1.0 / den.astype(np.float64))).astype(np.float32)
    assert np.array_equal(q1.view(np.uint32), q2.view(np.uint32))


def test_greedy_division_fma_form():
    """csrc/cd.hpp::greedy_div_fma (round 6, the register form of GreedyCD's sweep): q0 = RN(g r), e = fma(-den, q0, g), q = fma(e, r, q0)
    with r = RN(1 / den) is the correctly rounded g / den.  Every step evaluated in exact rational arithmetic and rounded once, as the
    hardware's fused multiply-add does; operands: wide-range random ones, divisors whose significand is all ones (the worst case for a
    reciprocal), small-integer pairs (what the bit-exact device tests feed), quotients next to a rounding boundary."""
    import math
    import struct
    from fractions import Fraction

    def rn32(fr):
        if fr == 0:
            return np.float32(0)
        sgn = -1 if fr < 0 else 1
        fr = abs(fr)
        e = math.floor(math.log2(fr))
        while Fraction(2) ** e > fr:
            e -= 1
        while Fraction(2) ** (e + 1) <= fr:
            e += 1
        scaled = fr / Fraction(2) ** (e - 23)
        n = scaled.numerator // scaled.denominator
        rem = scaled - n
        if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (n & 1)):
            n += 1
        return np.float32(sgn * float(n) * 2.0 ** (e - 23))

    def check(g, den):
        r = np.float32(np.float32(1) / den)
        q0 = np.float32(g * r)
        e = rn32(Fraction(float(g)) - Fraction(float(den)) * Fraction(float(q0)))
        q = rn32(Fraction(float(q0)) + Fraction(float(e)) * Fraction(float(r)))
        return q == rn32(Fraction(float(g)) / Fraction(float(den)))

    rng = np.random.default_rng(3)
    pairs = []
    for _ in range(6000):
        pairs.append((np.float32(rng.standard_normal() * 10 ** rng.uniform(-6, 6)), np.float32(abs(rng.standard_normal()) * 10 ** rng.uniform(-6, 6) + 1e-9)))
    for m in range(2000):
        den = np.frombuffer(struct.pack("<I", 0x3F800000 | (0x7FFFFF - (m % 64))), dtype=np.float32)[0]
        pairs.append((np.float32(1 + m % 97) * np.float32(1 + 2.0 ** -(m % 23)), den))
    for _ in range(4000):
        pairs.append((np.float32(rng.integers(-(1 << 20), 1 << 20)), np.float32(rng.integers(1, 1 << 12))))
    for _ in range(3000):      # g = the float nearest to (midpoint of two floats) * den
        den = np.float32(1 + rng.random())
        qf = np.float32(1 + rng.random())
        pairs.append((np.float32((float(qf) + float(np.spacing(qf)) / 2) * float(den)), den))
    bad = [(g, d) for g, d in pairs if g != 0 and not check(g, d)]
    assert not bad, bad[:5]


def test_alspgrad_gradient_option_validation():
    with pytest.raises(ValueError):
        nmfx.ALSPGrad(np.float32, gradient="fast")
    with pytest.raises(ValueError):
        nmfx.ALSPGrad(np.float32, gradient=0)
    assert nmfx.ALSPGrad(np.float32, gradient="exact")._opts()["pg_refresh"] == 1
    assert nmfx.ALSPGrad(np.float32, gradient=32)._opts()["pg_refresh"] == 32
    assert nmfx.ALSPGrad(np.float32)._opts()["pg_refresh"] == 0
