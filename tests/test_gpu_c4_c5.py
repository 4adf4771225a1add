"""Parity at the BASELINE.json workloads C4 and C5, on ONE MI355X (both fit: X is 8 GiB).

C4: X = 16384 x 131072, k = 256, fp32, alg = :projals       (src/projals.jl:76-107)
C5: X = 32768 x 32768,  k = 512, fp64, alg = :alspgrad      (src/alspgrad.jl:86-191, 242-347, 400-425), maxsubiter = 200

The CPU oracle cannot finish these sizes in seconds (one projals iteration is 2.2 TFLOP, one alspgrad outer iteration up to
~800 k x k x n products), so -- as for C3 in test_gpu_fullsize.py -- an iteration is checked through size-independent
pieces:

  projals   H_new[:, J] = max(0, (W'W + lh I)^-1 W'X[:, J])  depends only on W and the sampled columns of X    -> fp64 NumPy
            W_new[I, :] = max(0, X[I, :] H' (HH' + lw I)^-1)  depends only on H_new and the sampled rows of X   -> fp64 NumPy
            objective   = 0.5 ||X - WH||^2 + 0.5 lw ||W||^2 + 0.5 lh ||H||^2    against a blocked fp64 evaluation on the device
  alspgrad  the sub-solvers see X only through B = W'X (resp. XH') and the k x k Gram.  Those two are formed in fp64 by an
            independent implementation (torch / rocBLAS -- plumbing, like the synthetic X itself), and the ORACLE's
            `_pgrad_subsolve` (oracle/nmf_oracle.py, the restatement of alspgrad.jl:86-191 / :242-347) then runs the same
            sub-solve on the host, capped at a few inner iterations so that it finishes in seconds: the trajectory is
            deterministic, so equality of the first inner iterations (iterate, inner and back-track counts) is parity of the
            sub-solver at the full size.  The full maxsubiter = 200 outer iteration is then checked through its properties
            (objective against fp64, monotone decrease, caps on the counters, non-negativity).

X never exists on the host (8 GiB): it is generated on the device and handed over with nmfx_set_X_device.
"""
import numpy as np
import pytest
import torch

import nmf_oracle as orc
import nmfx

pytestmark = pytest.mark.gpu


def _device_problem(p, n, k, tdtype, seed, normalize_w0, warm=0.0):
    """Planted-rank X = Wg Hg + 0.01 U on the device as Xt (n x p row-major == X column-major), host W0, H0.
    warm > 0: W0 = Wg + warm * U (a warm start near the planted factor, like init = :custom / an SVD-based init) instead of U."""
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    Wg = torch.rand((p, k), generator=g, device=dev, dtype=tdtype)
    Hg = torch.rand((k, n), generator=g, device=dev, dtype=tdtype)
    Xt = torch.empty((n, p), device=dev, dtype=tdtype)
    blk = 8192
    for j0 in range(0, n, blk):
        j1 = min(n, j0 + blk)
        torch.matmul(Hg[:, j0:j1].t(), Wg.t(), out=Xt[j0:j1])
        Xt[j0:j1].add_(torch.rand((j1 - j0, p), generator=g, device=dev, dtype=tdtype), alpha=0.01)
    npdt = np.float32 if tdtype == torch.float32 else np.float64
    rng = np.random.default_rng(seed)
    W0 = rng.random((p, k))
    if warm > 0:
        W0 = Wg.cpu().numpy().astype(np.float64) + warm * W0
    if normalize_w0:
        W0 /= W0.sum(axis=0, keepdims=True)
    H0 = rng.random((k, n))
    return Xt, np.asfortranarray(W0.astype(npdt)), np.asfortranarray(H0.astype(npdt))


def _objective_fp64(Xt, W, H, blk=8192):
    """0.5 * sum (X - WH)^2 with the product and the sum in fp64, on the device, in column blocks of X."""
    dev = Xt.device
    Wd = torch.from_numpy(np.ascontiguousarray(W)).to(dev, torch.float64)
    Hd = torch.from_numpy(np.ascontiguousarray(H)).to(dev, torch.float64)
    s = 0.0
    for j0 in range(0, Xt.shape[0], blk):
        j1 = min(Xt.shape[0], j0 + blk)
        d = Xt[j0:j1].to(torch.float64) - Hd[:, j0:j1].t() @ Wd.t()
        s += float((d * d).sum())
    return 0.5 * s


def test_c4_projals_one_iteration_pieces(built):
    T = np.float32
    p, n, k = 16384, 131072, 256
    # Start: W0 = Wg + 0.05 U (warm start, not column-normalised).  From a cold U[0,1) start the first H = (W'W + lambda I)^-1 W'X
    # is M*Hg with cond(M) ~ 1e3, so cond(HH' + lambda I) reaches 1e8 (measured) -- beyond fp32 for ANY implementation of
    # pdrsolve! (the reference's potrf! / potri! included), and a column-normalised cold W0 is not even numerically positive
    # definite.  The warm start keeps both Grams at cond ~ 1e3, where an fp32 solve has a meaningful answer to compare.
    Xt, W0, H0 = _device_problem(p, n, k, torch.float32, 131072, normalize_w0=False, warm=0.05)
    H0[:] = 0                                                       # nnmf passes H = 0 to projals (src/interf.jl:39,43)
    lam = float(T(np.cbrt(np.finfo(T).eps)))                        # ProjectedALS defaults (src/projals.jl:30-31)
    with nmfx.Context(T, p, n, k) as ctx:
        ctx.set_X_device(Xt.data_ptr(), p)
        W1, H1 = W0.copy(order="F"), H0.copy(order="F")
        o = nmfx.make_opts(T, maxiter=1, tol=1e-30, lambda_w=lam, lambda_h=lam, track_objective=True)
        res, trace = ctx.solve(nmfx._lib.ALG_PROJALS, o, W1, H1)
        assert res.niters == 1 and res.status == 0
        # --- H update on a column sample: pdsolve! then projectnn! (src/projals.jl:92-95, src/utils.jl:63-70)
        J = np.sort(np.random.default_rng(0).choice(n, 96, replace=False))
        XJ = Xt[torch.from_numpy(J).to(Xt.device)].cpu().numpy().T.astype(np.float64)     # p x |J|
        W64 = W0.astype(np.float64)
        A = W64.T @ W64 + lam * np.eye(k)
        Href = np.maximum(np.linalg.solve(A, W64.T @ XJ), 0.0)
        errH = np.max(np.abs(H1[:, J] - Href)) / np.max(np.abs(Href))
        # --- W update on a row sample, from the NEW H: pdrsolve! then projectnn! (src/projals.jl:100-103, src/utils.jl:72-84)
        I = np.sort(np.random.default_rng(1).choice(p, 64, replace=False))
        XI = Xt[:, torch.from_numpy(I).to(Xt.device)].cpu().numpy().T.astype(np.float64)  # |I| x n
        H64 = H1.astype(np.float64)
        B = H64 @ H64.T + lam * np.eye(k)
        Wref = np.maximum(np.linalg.solve(B, (XI @ H64.T).T).T, 0.0)
        errW = np.max(np.abs(W1[I, :] - Wref)) / np.max(np.abs(Wref))
        condA, condB = np.linalg.cond(A), np.linalg.cond(B)
        print(f"[C4] errH={errH:.2e} (cond {condA:.2e})  errW={errW:.2e} (cond {condB:.2e})")
        # fp32 against fp64.  The right-hand sides W'X / XH' are fp32 sums of p resp. n products (relative rounding error
        # ~ eps * sqrt(length) at best), and the solve amplifies input perturbations by cond(Gram): the bound is
        # cond * eps * sqrt(contraction length) -- the error ANY fp32 pdsolve!/pdrsolve! (LAPACK's included) is entitled to
        eps = np.finfo(T).eps
        assert errH <= condA * eps * np.sqrt(p)
        assert errW <= condB * eps * np.sqrt(n)
        # --- objective (src/projals.jl:65-74) against an fp64 evaluation
        ref = _objective_fp64(Xt, W1, H1) + 0.5 * lam * float(np.sum(W1.astype(np.float64) ** 2)) \
            + 0.5 * lam * float(np.sum(H1.astype(np.float64) ** 2))
        assert abs(trace[1] - ref) <= 1e-5 * ref
        assert np.all(W1 >= 0) and np.all(H1 >= 0) and np.isfinite(W1).all() and np.isfinite(H1).all()
        # --- a few more iterations: the regularised objective keeps decreasing on this planted problem, and the
        #     stand-alone objective pass agrees with the tracked one
        o = nmfx.make_opts(T, maxiter=3, tol=1e-30, lambda_w=lam, lambda_h=lam, track_objective=True)
        res, tr2 = ctx.solve(nmfx._lib.ALG_PROJALS, o, W1, H1)
        assert res.niters == 3
        print("[C4] objective trajectory", trace[:2], tr2[:4])
        assert abs(tr2[0] - trace[1]) <= 1e-6 * trace[1]
        assert np.all(np.diff(tr2[:4]) <= 1e-6 * tr2[0])
        ref3 = _objective_fp64(Xt, W1, H1) + 0.5 * lam * float(np.sum(W1.astype(np.float64) ** 2)) \
            + 0.5 * lam * float(np.sum(H1.astype(np.float64) ** 2))
        assert abs(tr2[3] - ref3) <= 1e-5 * ref3


def _subproblem_value(Gram, B, Z, left):
    """f(Z) = 0.5 <Z, Gram Z> - <B, Z>  (the quadratic the projected-gradient sub-solver decreases)."""
    GZ = Gram @ Z if left else Z @ Gram
    return 0.5 * float(np.vdot(GZ, Z)) - float(np.vdot(B, Z))


def test_c5_alspgrad_subsolvers_and_outer_iteration(built):
    T = np.float64
    p = n = 32768
    k = 512
    Xt, W0, H0 = _device_problem(p, n, k, torch.float64, 32768, normalize_w0=True)
    dev = Xt.device
    tolg = float(np.finfo(T).eps ** 0.25)                            # ALSPGrad default (src/alspgrad.jl:363)
    with nmfx.Context(T, p, n, k) as ctx:
        ctx.set_X_device(Xt.data_ptr(), p)
        # ---- H sub-solver (alspgrad_updateh!, src/alspgrad.jl:69-191) at full size vs the oracle, first inner iterations
        CAP = 5
        Wd = torch.from_numpy(W0).to(dev)
        WtW = (Wd.t() @ Wd).cpu().numpy()
        WtX = np.asfortranarray((Xt @ Wd).t().cpu().numpy())        # k x n
        Hc = H0.copy(order="F")
        cnt = {"inner": 0, "backtracks": 0}
        t_or = orc._pgrad_subsolve(Hc, WtW, WtX, True, CAP, 20, T(tolg), 0.2, 0.01, T, cnt)
        Hg, Wg = H0.copy(order="F"), W0.copy(order="F")
        r = ctx.subsolve(0, nmfx.make_opts(T, maxsubiter=CAP, tolg=tolg), Wg, Hg)
        assert r.niters == t_or == CAP
        assert r.backtracks == cnt["backtracks"], (r.backtracks, cnt)
        assert np.max(np.abs(Hg - Hc)) <= 1e-9 * np.max(np.abs(Hc))
        assert np.array_equal(Wg, W0)
        f0, f5 = _subproblem_value(WtW, WtX, H0, True), _subproblem_value(WtW, WtX, Hg, True)
        assert f5 < f0
        # ---- W sub-solver (alspgrad_updatew!, :225-347) from the H above
        Hd = torch.from_numpy(Hg).to(dev)
        HHt = (Hd @ Hd.t()).cpu().numpy()
        XHt = np.asfortranarray((Hd @ Xt).t().cpu().numpy())        # p x k  (X H' = (H X')')
        Wc = W0.copy(order="F")
        cnt = {"inner": 0, "backtracks": 0}
        t_or = orc._pgrad_subsolve(Wc, HHt, XHt, False, CAP, 20, T(tolg), 0.2, 0.01, T, cnt)
        Wg2, Hg2 = W0.copy(order="F"), Hg.copy(order="F")
        r = ctx.subsolve(1, nmfx.make_opts(T, maxsubiter=CAP, tolg=tolg), Wg2, Hg2)
        assert r.niters == t_or == CAP
        assert r.backtracks == cnt["backtracks"], (r.backtracks, cnt)
        assert np.max(np.abs(Wg2 - Wc)) <= 1e-9 * np.max(np.abs(Wc))
        assert np.array_equal(Hg2, Hg)
        # ---- one full outer iteration with the reference defaults (maxsubiter = 200): update_wh!(::ALSPGradUpd), :400-425
        W1, H1 = W0.copy(order="F"), H0.copy(order="F")
        o = nmfx.make_opts(T, maxiter=2, maxsubiter=200, tolg=tolg, tol=1e-30, track_objective=True)
        res, trace = ctx.solve(nmfx._lib.ALG_ALSPGRAD, o, W1, H1)
        assert res.niters == 2 and res.status == 0
        print(f"[C5] 2 outer iterations: inner={res.inner_iters} backtracks={res.backtracks} loop={res.seconds_loop:.2f}s "
              f"objective {trace[0]:.6e} -> {trace[1]:.6e} -> {trace[2]:.6e}")
        assert 4 <= res.inner_iters <= 4 * 200
        assert res.inner_iters <= res.backtracks + 4 <= 20 * res.inner_iters + 4
        assert trace[2] < trace[1] < trace[0]
        ref = _objective_fp64(Xt, W1, H1)
        assert abs(trace[2] - ref) <= 1e-10 * ref
        assert np.all(W1 >= 0) and np.all(H1 >= 0) and np.isfinite(W1).all() and np.isfinite(H1).all()


@pytest.mark.parametrize("shape", [(1024, 1024, 16), (2048, 2048, 32)])
def test_c5_aspect_downscaled_counters(built, shape):
    """Same aspect (p = n = 64 k) as C5, fp64, reference defaults incl. maxsubiter = 200: the inner-iteration and back-track
    counters and the objective trajectory equal the oracle's over whole outer iterations."""
    from problems import planted, rel_trace_err
    T = np.float64
    p, n, k = shape
    X, W0, H0 = planted(p, n, k, T, seed=5 + k)
    alg = nmfx.ALSPGrad(T, maxiter=3, tol=1e-30)
    assert alg.maxsubiter == 200
    Wg, Hg = W0.copy(order="F"), H0.copy(order="F")
    r = nmfx.solve(alg, X, Wg, Hg, track_objective=True)
    Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
    ro = orc.solve("alspgrad", X, Wc, Hc, orc.Opts(maxiter=3, tol=1e-30, track_objective=True))
    assert r.niters == ro.niters == 3
    assert r.info["inner_iters"] == ro.counters["inner"]
    assert r.info["backtracks"] == ro.counters["backtracks"]
    assert rel_trace_err(r.trace, ro.trace) < 1e-9
    assert np.max(np.abs(Wg - Wc)) <= 1e-7 * np.max(np.abs(Wc))


def test_c4_projals_cold_start_h_solve_is_exact_at_full_size(built):
    """C4 at full size from a COLD start on which the answer is known exactly (no warm start, no sampling): W0 = indicators of
    disjoint row triples, lambda = 1, integer X generated on the device.  W'W + lambda I = 4 I, so the first H-side solve is
    H = max(0, W'X / 4) in exact arithmetic AND in Float32 (sums of three small integers, divisions by powers of two): all
    256 x 131072 entries of the device's H -- Gram launch, split-K product of 16384 x 131072 x 256 with its slab combine, blocked
    potrf, triangular inverse, two solve products with the clamp -- must equal the closed form bit for bit."""
    T = np.float32
    p, n, k = 16384, 131072, 256
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    Xt = torch.randint(0, 4, (n, p), generator=g, device=dev, dtype=torch.int32).to(torch.float32)   # n x p row-major == X column-major
    W0 = np.zeros((p, k), dtype=T, order="F")
    for j in range(k):
        W0[3 * j:3 * j + 3, j] = 1
    H0 = np.zeros((k, n), dtype=T, order="F")
    expect_t = (Xt[:, :3 * k].reshape(n, k, 3).sum(dim=2) / 4.0).cpu().numpy()                        # H' (n x k)
    o = nmfx.make_opts(T, maxiter=1, tol=1e-30, lambda_w=1.0, lambda_h=1.0, check_every=1000)
    H = np.empty((k, n), dtype=T, order="F")
    with nmfx.Context(T, p, n, k) as ctx:
        ctx.set_X_device(Xt.data_ptr(), p)
        ctx.set_factors(W0, H0)
        res, _ = ctx.iterate(2, o)
        ctx.get_factors(None, H)
    assert res.niters == 1 and np.isfinite(res.objvalue)
    assert np.array_equal(H.T, expect_t)
    assert float(H.max()) > 0


def test_c5_alspgrad_first_projected_gradient_step_is_bit_identical_at_full_size(built):
    """C5 at full size (32768 x 32768, k = 512, f64) on small-integer X, W, H: W'W, W'X and the gradient G = W'W H - W'X are exact
    in Float64, so the first inner iteration of alspgrad_updateh! (resp. updatew!) -- trial points max(Z - alpha G, 0), the
    back-tracking on sums whose last bits may differ, the accepted step -- must leave the factor BIT-IDENTICAL to the oracle's
    `_pgrad_subsolve` run on the exact Gram and B (src/alspgrad.jl:86-191, 242-347)."""
    T = np.float64
    p = n = 32768
    k = 512
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    Xt = torch.randint(0, 4, (n, p), generator=g, device=dev, dtype=torch.int32).to(torch.float64)     # n x p row-major == X column-major
    rng = np.random.default_rng(5)
    W0 = np.asfortranarray(rng.integers(0, 3, size=(p, k)).astype(T))
    H0 = np.asfortranarray(rng.integers(0, 3, size=(k, n)).astype(T))
    tolg = 1e-30
    with nmfx.Context(T, p, n, k) as ctx:
        ctx.set_X_device(Xt.data_ptr(), p)
        Wd, Hd = torch.from_numpy(W0).to(dev), torch.from_numpy(H0).to(dev)
        WtW = (Wd.t() @ Wd).cpu().numpy()
        WtX = np.asfortranarray((Xt @ Wd).t().cpu().numpy())        # k x n
        HHt = (Hd @ Hd.t()).cpu().numpy()
        XHt = np.asfortranarray((Hd @ Xt).t().cpu().numpy())        # p x k
        for a in (WtW, WtX, HHt, XHt):
            assert np.array_equal(a, np.rint(a)) and np.abs(a).max() < 2 ** 50
        for side, Gram, B, Z0 in ((0, WtW, WtX, H0), (1, HHt, XHt, W0)):
            Zc = Z0.copy(order="F")
            cnt = {"inner": 0, "backtracks": 0}
            t_or = orc._pgrad_subsolve(Zc, Gram, B, side == 0, 1, 20, T(tolg), 0.2, 0.01, T, cnt)
            Wg, Hg = W0.copy(order="F"), H0.copy(order="F")
            r = ctx.subsolve(side, nmfx.make_opts(T, maxsubiter=1, tolg=tolg), Wg, Hg)
            got = Hg if side == 0 else Wg
            assert r.niters == t_or == 1 and r.backtracks == cnt["backtracks"], (r.niters, t_or, r.backtracks, cnt)
            assert np.array_equal(got.view(np.uint64), Zc.view(np.uint64)), float(np.max(np.abs(got - Zc)))
            assert not np.array_equal(got, Z0)
