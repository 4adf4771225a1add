"""Parity at the BASELINE.json sizes.

C2 (4096 x 4096, k=64, f32): the NumPy oracle still finishes in seconds, so the objective trajectory is compared
iteration for iteration like in the small tests.
C3 (16384 x 16384, k=256, f32): the oracle needs ~10 s per iteration, so one iteration is checked through
size-independent pieces of the update rule instead (SURVEY.md section 8c "size-independent properties"):
  * H-update columns: H_new[:, J] depends only on X[:, J], H[:, J] and W  -> fp64 NumPy on a column sample
  * W-update rows:    W_new[I, :] depends only on X[I, :], W[I, :] and H_new -> fp64 NumPy on a row sample
  * the objective of the final factors against a full fp64-accumulated NumPy evaluation
  * monotone non-increasing objective, non-negativity, zero-preservation (multiplicative updates keep exact zeros)
"""
import numpy as np
import pytest
import torch

import nmf_oracle as orc
import nmfx
from problems import planted, rel_trace_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("obj", ["mse", "div"])
def test_c2_trajectory(built, obj):
    T = np.float32
    p = n = 4096
    k = 64
    X, W0, H0 = planted(p, n, k, T, seed=4096)
    alg = nmfx.MultUpdate(T, obj=obj, maxiter=6, tol=1e-30)
    Wg, Hg = W0.copy(order="F"), H0.copy(order="F")
    r = nmfx.solve(alg, X, Wg, Hg, track_objective=True)
    Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
    ro = orc.solve("mult" + obj, X, Wc, Hc, orc.Opts(maxiter=6, tol=1e-30, track_objective=True))
    assert r.niters == ro.niters == 6
    assert rel_trace_err(r.trace, ro.trace) < 1e-5          # north_star: objective within 1e-5 relative
    assert np.max(np.abs(Wg - Wc)) <= 1e-3 * np.max(np.abs(Wc))
    assert np.max(np.abs(Hg - Hc)) <= 1e-3 * np.max(np.abs(Hc))


import functools


@functools.lru_cache(maxsize=1)
def _c3_inputs():
    """(X, W0, H0) of the headline shape; built once per session (callers copy W0 / H0 and never write to X)."""
    p = n = 16384
    k = 256
    g = torch.Generator(device="cuda")
    g.manual_seed(16384)
    Wg = torch.rand((p, k), generator=g, device="cuda")
    Hg = torch.rand((k, n), generator=g, device="cuda")
    Xt = Hg.t().contiguous() @ Wg.t().contiguous()                      # (n, p) row-major == X column-major
    Xt.add_(torch.rand(Xt.shape, generator=g, device="cuda"), alpha=0.01)
    X = np.asfortranarray(Xt.cpu().numpy().T)
    rng = np.random.default_rng(3)
    W0 = rng.random((p, k), dtype=np.float32)
    W0 /= W0.sum(axis=0, keepdims=True)
    H0 = rng.random((k, n), dtype=np.float32)
    W0[::97, 3] = 0.0                                                    # exact zeros must survive multiplicative updates
    H0[5, ::101] = 0.0
    return X, np.asfortranarray(W0), np.asfortranarray(H0)


def test_c3_multmse_one_iteration_pieces(built):
    T = np.float32
    X, W0, H0 = _c3_inputs()
    p, n = X.shape
    k = W0.shape[1]
    delta = float(T(np.sqrt(np.finfo(T).eps)))
    with nmfx.Context(T, p, n, k) as ctx:
        ctx.set_X(X)
        W1, H1 = W0.copy(order="F"), H0.copy(order="F")
        o = nmfx.make_opts(T, maxiter=1, tol=1e-30, track_objective=True)
        res, trace = ctx.solve(0, o, W1, H1)
        assert res.niters == 1
        # --- H update on a column sample (fp64 reference of src/multupd.jl:98-103 in Gram form)
        J = np.random.default_rng(0).choice(n, 96, replace=False)
        W64 = W0.astype(np.float64)
        num = W64.T @ X[:, J].astype(np.float64)
        den = (W64.T @ W64) @ H0[:, J].astype(np.float64) + delta
        Href = H0[:, J] * (np.maximum(num, 0) / den)
        assert np.max(np.abs(H1[:, J] - Href)) <= 2e-5 * np.max(np.abs(Href))
        # --- W update on a row sample (uses the NEW H, src/multupd.jl:109-114)
        I = np.random.default_rng(1).choice(p, 64, replace=False)
        H64 = H1.astype(np.float64)
        numw = X[I, :].astype(np.float64) @ H64.T
        denw = W0[I, :].astype(np.float64) @ (H64 @ H64.T) + delta
        Wref = W0[I, :] * (np.maximum(numw, 0) / denw)
        assert np.max(np.abs(W1[I, :] - Wref)) <= 2e-5 * np.max(np.abs(Wref))
        # --- objective of the result: fp32 product like the reference, Float64 accumulation (sqL2dist)
        ref_obj = 0.5 * orc.sqL2dist(X, W1 @ H1)
        assert abs(trace[1] - ref_obj) <= 1e-5 * ref_obj
        # --- properties
        assert np.all(W1 >= 0) and np.all(H1 >= 0) and np.isfinite(W1).all() and np.isfinite(H1).all()
        assert not W1[::97, 3].any() and not H1[5, ::101].any()
        # a few more iterations: Lee-Seung updates never increase the objective (lambda = 0)
        o = nmfx.make_opts(T, maxiter=4, tol=1e-30, track_objective=True)
        res, tr2 = ctx.solve(0, o, W1, H1)
        assert np.all(np.diff(tr2[: res.niters + 1]) <= 1e-6 * tr2[0])


@pytest.mark.parametrize("obj", ["mse", "div"])
def test_c3_trajectory_vs_oracle(built, obj):
    """north_star: "iteration-for-iteration objective value and final W, H" on the metric's OWN workload -- the CPU oracle
    (the reference's operation sequence: 6 / 4 products of 2pnk per iteration through p x n temporaries, src/multupd.jl:83-116,
    150-193; objective per iteration as with verbose = true, src/common.jl:76-82) runs three (MSE) / two (divergence) tracked iterations at
    X = 16384 x 16384, k = 256, Float32, the device the same: every point of the
    objective trajectory within 1e-5 relative, final factors within 1e-3 of max|.|."""
    T = np.float32
    X, W0, H0 = _c3_inputs()
    p, n = X.shape
    k = W0.shape[1]
    lam = float(T(np.sqrt(np.finfo(T).eps))) if obj == "div" else 0.0
    iters = 3 if obj == "mse" else 2        # (the divergence oracle's element-wise passes over 16384^2 cost ~50 s of host time per iteration)
    Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
    ro = orc.solve("mult" + obj, X, Wc, Hc, orc.Opts(maxiter=iters, tol=1e-30, lambda_w=lam, lambda_h=lam, track_objective=True))
    with nmfx.Context(T, p, n, k) as ctx:
        ctx.set_X(X)
        Wg, Hg = W0.copy(order="F"), H0.copy(order="F")
        o = nmfx.make_opts(T, maxiter=iters, tol=1e-30, lambda_w=lam, lambda_h=lam, track_objective=True)
        res, trace = ctx.solve(0 if obj == "mse" else 1, o, Wg, Hg)
    assert res.niters == ro.niters == iters
    err = np.abs(np.asarray(trace[:iters + 1]) - np.asarray(ro.trace)) / np.abs(np.asarray(ro.trace))
    print(f"C3 mult{obj}: relative objective error per point {err}")
    assert np.all(err < 1e-5), err
    assert np.max(np.abs(Wg - Wc)) <= 1e-3 * np.max(np.abs(Wc))
    assert np.max(np.abs(Hg - Hc)) <= 1e-3 * np.max(np.abs(Hc))
    assert not Wg[::97, 3].any() and not Hg[5, ::101].any() and not Wc[::97, 3].any() and not Hc[5, ::101].any()


def test_c3_multdiv_properties(built):
    T = np.float32
    X, W0, H0 = _c3_inputs()
    p, n = X.shape
    k = W0.shape[1]
    eps = np.finfo(T).eps
    lam = float(T(np.sqrt(eps)))
    with nmfx.Context(T, p, n, k) as ctx:
        ctx.set_X(X)
        W1, H1 = W0.copy(order="F"), H0.copy(order="F")
        o = nmfx.make_opts(T, maxiter=3, tol=1e-30, track_objective=True, lambda_w=lam, lambda_h=lam)
        res, trace = ctx.solve(1, o, W1, H1)
        assert res.niters == 3
        assert np.all(np.diff(trace[:4]) <= 1e-6 * trace[0])           # KL multiplicative updates are monotone too
        assert np.all(W1 >= 0) and np.all(H1 >= 0) and np.isfinite(W1).all() and np.isfinite(H1).all()
        assert not W1[::97, 3].any() and not H1[5, ::101].any()
        # generalized KL divergence of the final factors (gkldiv, term in T, Float64 accumulation)
        ref = orc.gkldiv(X, W1 @ H1)            # (in column blocks: the one-shot form's 1 GiB temporaries cost ~25 s of page faults here)
        assert abs(trace[3] - ref) <= 2e-5 * abs(ref)


def _cd_row_sweep(Wrows, P, Z):
    """fp64 restatement of the CD sweep (coorddesc.jl:133-156) for a SAMPLE of rows: rows do not interact."""
    W = Wrows.copy()
    for t in range(P.shape[0]):
        grad = W @ P[t] - Z[:, t]
        if P[t, t] != 0:
            W[:, t] = np.maximum(W[:, t] - grad / P[t, t], 0.0)
    return W


@pytest.mark.parametrize("alg", ["cd", "greedycd"])
def test_c2_coordinate_descent_pieces(built, alg):
    """C2 size (4096 x 4096, k = 64), f64.  Size-independent pieces of one outer iteration: the sweep of a sample row depends
    only on that row of W (or column of H), the k x k Gram and the row of the numerator (plus, for GreedyCD, the global
    scalar p_init, which a vectorised NumPy pass over all rows provides)."""
    T = np.float64
    p = n = 4096
    k = 64
    X, W0, H0 = planted(p, n, k, T, seed=77)
    inst = nmfx.CoordinateDescent(T, maxiter=2, tol=1e-30) if alg == "cd" else nmfx.GreedyCD(T, maxiter=2, tol=1e-30)
    algid = inst._alg()
    with nmfx.Context(T, p, n, k) as ctx:
        ctx.set_X(X)
        W1, H1 = W0.copy(order="F"), H0.copy(order="F")
        res, trace = ctx.solve(algid, nmfx.make_opts(T, maxiter=1, tol=1e-30, track_objective=True), W1, H1)
        assert res.niters == 1
        I = np.random.default_rng(0).choice(p, 24, replace=False)
        J = np.random.default_rng(1).choice(n, 24, replace=False)
        HHt, XHt = H0 @ H0.T, X @ H0.T
        WtW, XtW = W1.T @ W1, X.T @ W1
        if alg == "cd":
            Wref = _cd_row_sweep(W0[I], HHt, XHt[I])
            Href = _cd_row_sweep(H0.T[J], WtW, XtW[J]).T
        else:
            eps = np.finfo(T).eps

            def sweep(Wv, P, Z, rows):
                G = Wv @ P - Z
                prr = np.diag(P)
                S = np.maximum(0.0, Wv - G / (eps + prr)) - Wv
                D = -G * S - (0.5 * prr) * (S * S)
                p_init = max(-1.0, D.max(axis=1).max())
                out = []
                for i in rows:
                    w, g = Wv[i].copy(), G[i].copy()
                    wn = np.zeros(k)
                    for _ in range(k * k):
                        s = np.maximum(0.0, w - g / (eps + prr)) - w
                        d = -g * s - (0.5 * prr) * (s * s)
                        q = int(np.argmax(d))
                        if d[q] < 0.001 * p_init:
                            break
                        wn[q] += s[q]
                        g = g + s[q] * P[q]
                    out.append(np.maximum(w + wn, 0.0))
                return np.array(out)

            Wref = sweep(W0, HHt, XHt, I)
            Href = sweep(np.ascontiguousarray(H0.T), WtW, XtW, J).T
        assert np.max(np.abs(W1[I] - Wref)) <= 1e-9 * max(1.0, np.max(np.abs(Wref)))
        assert np.max(np.abs(H1[:, J] - Href)) <= 1e-9 * max(1.0, np.max(np.abs(Href)))
        ref_obj = 0.5 * float(np.sum((X - W1 @ H1) ** 2))
        assert abs(trace[1] - ref_obj) <= 1e-10 * ref_obj
        assert np.all(W1 >= 0) and np.all(H1 >= 0)
        # exact coordinate minimisation never increases the objective
        res, tr2 = ctx.solve(algid, nmfx.make_opts(T, maxiter=4, tol=1e-30, track_objective=True), W1, H1)
        assert np.all(np.diff(tr2[: res.niters + 1]) <= 1e-12 * tr2[0])


def test_c3_updates_are_bit_identical_on_exact_inputs(built):
    """C3 at full size (16384 x 16384, k = 256, f32) on inputs whose GEMMs are exact: X in {0..3}, W0 and H0 in {0, 1}.  Then
    W'X <= 49152, W'W <= 16384 and (W'W)H <= 2^22 are exact in Float32 in any summation order (split-K slabs, Gram tail
    pieces, MFMA accumulation), and H after the first iteration must equal  H0 .* (W'X) ./ ((W'W)H0 .+ delta)  evaluated
    element-wise in Float32 from exact integer operands, BIT FOR BIT, for all 256 x 16384 entries; likewise W with
    update_H = false.  (src/multupd.jl:98-115; the operands are formed in Float64 by torch -- plumbing -- and checked to be
    integers.)"""
    T = np.float32
    p = n = 16384
    k = 256
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    Xt = torch.randint(0, 4, (n, p), generator=g, device=dev, dtype=torch.int32).to(torch.float32)     # n x p row-major == X column-major
    rng = np.random.default_rng(11)
    W0 = np.asfortranarray(rng.integers(0, 2, size=(p, k)).astype(T))
    H0 = np.asfortranarray(rng.integers(0, 2, size=(k, n)).astype(T))
    X64 = Xt.t().to(torch.float64)                                                                      # p x n view, fp64 copy (2 GiB)
    W64 = torch.from_numpy(np.ascontiguousarray(W0)).to(dev, torch.float64)
    H64 = torch.from_numpy(np.ascontiguousarray(H0)).to(dev, torch.float64)
    delta = T(np.sqrt(np.finfo(T).eps))
    def rule(z0, num64, den64):
        num, den = num64.cpu().numpy(), den64.cpu().numpy()
        assert np.array_equal(num, np.rint(num)) and np.array_equal(den, np.rint(den)) and den.max() < 2 ** 24 and num.max() < 2 ** 24
        return (z0 * (num.astype(T) / (den.astype(T) + delta))).astype(T)                              # one rounding per operation, in T
    expect_H = rule(H0, W64.t() @ X64, (W64.t() @ W64) @ H64)
    expect_W = rule(W0, X64 @ H64.t(), W64 @ (H64 @ H64.t()))
    del X64
    with nmfx.Context(T, p, n, k) as ctx:
        ctx.set_X_device(Xt.data_ptr(), p)
        for update_H, expect in ((True, expect_H), (False, expect_W)):
            ctx.set_factors(W0, H0)
            o = nmfx.make_opts(T, maxiter=1, tol=1e-30, update_H=update_H, check_every=1000)
            res, _ = ctx.iterate(0, o)
            W, H = np.empty((p, k), T, order="F"), np.empty((k, n), T, order="F")
            ctx.get_factors(W, H)
            assert res.niters == 1
            got = H if update_H else W
            assert np.array_equal(got.view(np.uint32), np.asfortranarray(expect).view(np.uint32)), float(np.max(np.abs(got - expect)))
            if not update_H:
                assert np.array_equal(H, H0)


@pytest.mark.parametrize("alg", ["multmse", "cd"])
def test_c3_big_products_one_block_per_cu_against_the_split_form(built, monkeypatch, alg):
    """C3 shape, f32: the output of each big product is exactly one 128 x 128 tile per CU, so MultUpdate-MSE and CoordinateDescent launch
    them UNSPLIT (solver_impl.hpp: iterate) -- no smaller test shape takes that branch.  One iteration from the same start with the
    branch on and off (NMFX_UNSPLIT=0, the 2-way split every other test shape runs and the oracle comparisons pin): the two differ by
    the summation order of 16384-term Float32 sums only."""
    X, W0, H0 = _c3_inputs()
    T = np.float32
    p, n = X.shape
    k = W0.shape[1]
    inst = {"multmse": nmfx.MultUpdate(T, obj="mse", maxiter=2, tol=1e-30), "cd": nmfx.CoordinateDescent(T, maxiter=2, tol=1e-30)}[alg]
    monkeypatch.setenv("NMFX_DEV", "1")
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("NMFX_UNSPLIT", mode)
        with nmfx.Context(T, p, n, k) as ctx:
            ctx.set_X(X)
            W1, H1 = W0.copy(order="F"), H0.copy(order="F")
            res, trace = ctx.solve(inst._alg(), nmfx.make_opts(T, maxiter=1, tol=1e-30, track_objective=True), W1, H1)
            assert res.niters == 1
            out[mode] = (W1, H1, trace[1])
    (Wa, Ha, oa), (Wb, Hb, ob) = out["1"], out["0"]
    dw = np.abs(Wa - Wb) / np.max(np.abs(Wb))
    dh = np.abs(Ha - Hb) / np.max(np.abs(Hb))
    assert abs(oa - ob) <= 1e-5 * abs(ob)
    assert dw.max() <= 1e-4          # (7.5e-6 / 9.1e-6 measured)
    if alg == "multmse":
        assert dh.max() <= 1e-4      # (6.9e-6 measured)
        return
    # The coordinate sweep amplifies the last bits of its inputs (a Gauss-Seidel pass over k = 256 coordinates against the Gram of a
    # random start: the two runs' W differ by 9e-6 and their H by up to 4 % in some columns -- and so do the Float64 sweeps evaluated from
    # the two W), so each form is compared with a Float64 evaluation of the sweep from ITS OWN W, on 8 columns of H at random plus the 8
    # in which the two runs differ most: 1.6e-4 (unsplit) and 7e-5 (split) measured.
    J = np.concatenate([np.random.default_rng(1).choice(n, 8, replace=False), np.argsort(np.max(np.abs(Ha - Hb), axis=0))[-8:]])
    for mode, (W1, H1, _) in out.items():
        W64 = W1.astype(np.float64)
        WtW, XtW = W64.T @ W64, X[:, J].astype(np.float64).T @ W64
        Href = _cd_row_sweep(H0.T[J].astype(np.float64), WtW, XtW).T
        assert np.max(np.abs(H1[:, J] - Href)) <= 1e-3 * np.max(np.abs(Href)), mode
