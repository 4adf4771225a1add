"""CPU ORACLE (NumPy twin) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Literal CPU restatement of the NMF.jl hot path (JuliaStats/NMF.jl v1.0.3):
the `nmf_skeleton!` driver and the `update_wh!` bodies of MultUpdate (MSE and
KL divergence), ProjectedALS and ALSPGrad, plus
CoordinateDescent with shuffle = false and GreedyCD.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
this module; the shipped solver (libnmfx.so) never does.

PARITY PINNING STATUS
  * Julia is not installed in the build container, so the reference itself
    cannot run here.  The oracle is pinned against every known-answer test the
    reference's own test-suite holds for this path (test/testproblems.jl:6-13,
    test/multupd.jl:3-22, test/alspgrad.jl:3-25, test/utils.jl:6-63,
    test/interf.jl:33-37, test/coorddesc.jl:5-8, test/greedycd.jl:5-20; the
    shuffle = true half of test/coorddesc.jl:10-14 needs Julia's RNG and is run
    in component order) -- see tests/test_oracle_kat.py -- and against an
    independent second restatement in C (oracle/nmf_oracle.c).
  * `Result.objvalue`: PARITY UNPINNED.  It is computed by StatsBase.sqL2dist /
    StatsBase.gkldiv (compat 0.25-0.34, not vendored, no Manifest; call sites
    src/multupd.jl:81,148, src/projals.jl:66, src/alspgrad.jl:398) and no
    reference test asserts its value.  Restated from the published StatsBase
    definitions: per-element term in T, running sum in Float64.
  * ProjectedALS: PARITY UNPINNED beyond the interface smoke run -- the
    reference has no projals test file (test/runtests.jl:9-16).

Conventions: X is p x n, W is p x k, H is k x n, all column-major (Fortran
order) like Julia `Matrix{T}`.  `mul!` -> BLAS gemm via `@`; sequential
T-precision accumulations of the Julia scalar loops are reproduced with
`np.cumsum(...)[-1]` (cumsum is a strict left-to-right recurrence in T).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
from scipy.linalg import lapack

MULTMSE, MULTDIV, PROJALS, ALSPGRAD, CD, GREEDYCD = 0, 1, 2, 3, 4, 5
ALG_NAMES = {"multmse": MULTMSE, "multdiv": MULTDIV, "projals": PROJALS, "alspgrad": ALSPGRAD, "cd": CD, "greedycd": GREEDYCD}


def eps(T):
    return float(np.finfo(T).eps)


@dataclass
class Opts:
    """Union of the option structs: MultUpdate (src/multupd.jl:9-42),
    ProjectedALS (src/projals.jl:18-34), ALSPGrad (src/alspgrad.jl:352-373)."""
    maxiter: int = 100
    tol: float = 0.0            # 0 -> cbrt(eps(T)) (struct default)
    update_H: bool = True
    lambda_w: float = -1.0      # <0 -> algorithm default
    lambda_h: float = -1.0
    delta: float = -1.0         # <0 -> sqrt(eps(T))          (multupd.jl:48,50)
    maxsubiter: int = 200       # alspgrad.jl:361
    tolg: float = -1.0          # <0 -> eps(T)^(1/4)          (alspgrad.jl:363)
    traceiter: int = 20         # alspgrad.jl:407
    beta: float = 0.2
    sigma: float = 0.01
    track_objective: bool = False   # verbose-style per-iteration objective (common.jl:76-82)
    # CoordinateDescentUpd's resolved regularisation (coorddesc.jl:62-82).  shuffle = true (coorddesc.jl:130-131) draws
    # randperm(k) from Julia's RNG per _update_coord_descent! call; the oracle takes the orders as an INPUT instead:
    # perm_source(call_index) -> permutation of range(k), call_index = 2*(t-1) + side (None: order 1..k, shuffle = false)
    perm_source: object = None
    l1_w: float = 0.0
    l2_w: float = 0.0
    l1_h: float = 0.0
    l2_h: float = 0.0


@dataclass
class Result:
    """NMF.Result{T} (src/common.jl:21-35)."""
    W: np.ndarray
    H: np.ndarray
    niters: int
    converged: bool
    objvalue: float
    trace: list = field(default_factory=list)   # objective at t=0..niters when tracked
    counters: dict = field(default_factory=dict)
    relchange: list = field(default_factory=list)   # stop_condition's devmax per iteration when tracked (entry 0 = NaN)


def resolve_opts(alg: int, T, o: Opts) -> Opts:
    """Fill dtype-dependent defaults exactly as the reference constructors do."""
    r = Opts(**o.__dict__)
    e = eps(T)
    if r.tol <= 0:
        r.tol = float(T(np.cbrt(e)))                      # multupd.jl:21, projals.jl:28, alspgrad.jl:362
    if alg in (MULTMSE, MULTDIV):
        if r.lambda_w < 0:
            r.lambda_w = 0.0                              # multupd.jl:23-24
        if r.lambda_h < 0:
            r.lambda_h = 0.0
        if alg == MULTDIV:                                # multupd.jl:37-40
            r.lambda_w = max(r.lambda_w, float(T(math.sqrt(e))))
            r.lambda_h = max(r.lambda_h, float(T(math.sqrt(e))))
    elif alg == GREEDYCD:
        if r.lambda_w < 0:
            r.lambda_w = 0.0                              # greedycd.jl:22-23
        if r.lambda_h < 0:
            r.lambda_h = 0.0
    elif alg == PROJALS:
        if r.lambda_w < 0:
            r.lambda_w = float(T(np.cbrt(e)))             # projals.jl:30-31
        if r.lambda_h < 0:
            r.lambda_h = float(T(np.cbrt(e)))
    else:
        r.lambda_w = r.lambda_h = 0.0
    if r.delta < 0:
        r.delta = float(T(math.sqrt(e)))                  # multupd.jl:48
    if r.tolg < 0:
        r.tolg = float(T(e ** 0.25))                      # alspgrad.jl:363
    return r


# ----------------------------------------------------------------------------
# StatsBase restatements (un-vendored dependency; see header)
# ----------------------------------------------------------------------------

_BLOCKED_ABOVE = 1 << 24   # elements: larger inputs are evaluated in column blocks (same terms, Float64 sum of the blocks' Float64 sums)


def _col_blocks(a, b):
    """Column blocks of two equally shaped 2-D arrays, ~4M elements each.  Only for the full-size parity runs (16384 x 16384): the
    whole-array form allocates half a dozen 1 GiB temporaries per call, and on the GPU box's micro-VM every fresh page costs a fault
    (0.25 GB/s): 30 s per objective evaluation against 3 s in blocks.  Small inputs keep the one-shot form (and its exact bits)."""
    n = a.shape[1]
    step = max(1, (1 << 22) // max(1, a.shape[0]))
    for j0 in range(0, n, step):
        yield a[:, j0:j0 + step], b[:, j0:j0 + step]


def _pmap(fn, blocks):
    """fn over the blocks on a small thread pool (NumPy's element-wise loops release the GIL), results in BLOCK ORDER: the sum of
    the blocks' sums does not depend on the schedule.  Full-size parity runs only (see _col_blocks)."""
    import concurrent.futures as cf
    import os
    blocks = list(blocks)
    with cf.ThreadPoolExecutor(max_workers=max(1, min(16, os.cpu_count() or 1))) as ex:
        return list(ex.map(lambda xy: fn(*xy), blocks))


def sqL2dist(a, b):
    """StatsBase.sqL2dist: r=0.0; r += abs2(a[i]-b[i]) -- term in T, sum in Float64."""
    if a.ndim == 2 and a.size > _BLOCKED_ABOVE:
        return float(sum(_pmap(sqL2dist, _col_blocks(a, b))))
    d = a - b
    return float(np.sum((d * d).astype(np.float64)))


def gkldiv(a, b):
    """StatsBase.gkldiv: sum(a>0 ? a*log(a/b) - a + b : b), term in T, sum in Float64."""
    if a.ndim == 2 and a.size > _BLOCKED_ABOVE:
        return float(sum(_pmap(gkldiv, _col_blocks(a, b))))
    T = a.dtype.type
    pos = a > 0
    safe_a = np.where(pos, a, T(1))
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(pos, safe_a * np.log(safe_a / b) - safe_a + b, b)
    return float(np.sum(t.astype(np.float64)))


# ----------------------------------------------------------------------------
# src/utils.jl
# ----------------------------------------------------------------------------

def adddiag(A, a):
    """utils.jl:15-24 (no-op when a == 0)."""
    if a != 0.0:
        idx = np.arange(A.shape[0])
        A[idx, idx] += A.dtype.type(a)
    return A


def projectnn(A):
    """utils.jl:34-41: entries < 0 become 0 (NaN passes through)."""
    A[A < 0] = 0
    return A


class PosDefException(Exception):
    pass


def _potrf(A):
    f = lapack.spotrf if A.dtype == np.float32 else lapack.dpotrf
    c, info = f(A, lower=0, clean=0, overwrite_a=0)
    if info != 0:
        raise PosDefException(f"potrf info={info}")
    return np.asfortranarray(c)


def pdsolve(A, x):
    """utils.jl:63-70: x <- inv(A) x via potrf!('U') + potrs!."""
    c = _potrf(A)
    f = lapack.spotrs if A.dtype == np.float32 else lapack.dpotrs
    sol, info = f(c, x, lower=0)
    assert info == 0
    return np.asfortranarray(sol.astype(A.dtype, copy=False))


def pdrsolve(A, B):
    """utils.jl:72-84: x <- A inv(B): potrf!, potri!, copytri!, then mul!."""
    c = _potrf(B)
    f = lapack.spotri if B.dtype == np.float32 else lapack.dpotri
    inv, info = f(c, lower=0)
    assert info == 0
    inv = np.triu(inv) + np.triu(inv, 1).T          # copytri!(B, 'U')
    return np.asfortranarray(A @ np.asfortranarray(inv))


# ----------------------------------------------------------------------------
# src/common.jl:92-111
# ----------------------------------------------------------------------------

def stop_condition(W, preW, H, preH, tol):
    """Component-wise relative change test, T-precision sequential sums.
    Early exit at the first failing component == all components pass."""
    T = W.dtype.type
    tol = T(tol)
    dw = np.cumsum((W - preW) ** 2, axis=0, dtype=W.dtype)[-1, :]
    sw = np.cumsum((W + preW) ** 2, axis=0, dtype=W.dtype)[-1, :]
    dh = np.cumsum((H - preH) ** 2, axis=1, dtype=H.dtype)[:, -1]
    sh = np.cumsum((H + preH) ** 2, axis=1, dtype=H.dtype)[:, -1]
    bad = (np.sqrt(dw) > tol * np.sqrt(sw)) | (np.sqrt(dh) > tol * np.sqrt(sh))
    return not bool(np.any(bad))


def stop_condition_dev(W, preW, H, preH, tol):
    """(converged, devmax) exactly as src/common.jl:92-111 returns them: devmax is the running maximum of
    sqrt(max(dev_w/sum_w, dev_h/sum_h)) over the components visited before the early `return false`."""
    T = W.dtype.type
    tol = T(tol)
    dw = np.cumsum((W - preW) ** 2, axis=0, dtype=W.dtype)[-1, :]
    sw = np.cumsum((W + preW) ** 2, axis=0, dtype=W.dtype)[-1, :]
    dh = np.cumsum((H - preH) ** 2, axis=1, dtype=H.dtype)[:, -1]
    sh = np.cumsum((H + preH) ** 2, axis=1, dtype=H.dtype)[:, -1]
    bad = (np.sqrt(dw) > tol * np.sqrt(sw)) | (np.sqrt(dh) > tol * np.sqrt(sh))
    jf = int(np.argmax(bad)) if bad.any() else len(bad) - 1
    with np.errstate(divide="ignore", invalid="ignore"):
        vals = np.sqrt(np.maximum(dw / sw, dh / sh))[: jf + 1]
    devmax = T(0)
    for v in vals:
        devmax = max(devmax, v)                                   # Julia max(x, NaN) = NaN; Python max keeps the first: fine for finite data
    return (not bool(bad.any())), float(devmax)


# ----------------------------------------------------------------------------
# updaters
# ----------------------------------------------------------------------------

class _MultMSE:
    """src/multupd.jl:56-116."""

    def __init__(self, T, o, X, W, H):
        self.T, self.o = T, o
        self.WH = W @ H                                           # :72

    def objv(self, X, W, H):
        return float(self.T(0.5 * sqL2dist(X, self.WH)))          # :81, rounded to T by Result{T}

    def update(self, X, W, H):
        T, o = self.T, self.o
        lw, lh, d = T(o.lambda_w), T(o.lambda_h), T(o.delta)
        if o.update_H:
            WtX = W.T @ X                                         # :98
            WtWH = W.T @ self.WH                                  # :99
            H *= np.maximum(T(0), WtX - lh) / (WtWH + d)          # :101-103
            self.WH = W @ H                                       # :104
        XHt = X @ H.T                                             # :109
        WHHt = self.WH @ H.T                                      # :110
        W *= np.maximum(T(0), XHt - lw) / (WHHt + d)              # :112-114
        self.WH = W @ H                                           # :115


class _MultMSEState:
    """MultUpdMSE_State (src/multupd.jl:63-80) as the reference keeps it: WH, WtX, WtWH, XHt, WHHt allocated ONCE in
    prepare_state, every mul! and every element-wise loop of update_wh! (:83-116) writing in place.  Same operations in the
    same order as _MultMSE.update (the factors agree to a few ulp, tests/test_oracle_kat.py: `matmul(..., out=F-ordered)` takes
    another gemm call form than `@`, nothing else differs) but no temporaries: this is the form
    bench.py's cpu_baseline times (the allocating form spends most of its time in first-touch page faults of the p x n
    temporaries, which NMF.jl does not have).  `phases` (optional dict) accumulates seconds per call site."""

    def __init__(self, T, o, X, W, H):
        self.T, self.o = T, o
        p, n = X.shape
        k = W.shape[1]
        f = dict(dtype=T, order="F")
        self.WH = np.empty((p, n), **f)
        np.matmul(W, H, out=self.WH)                              # :72
        self.WtX, self.WtWH, self.tH = (np.empty((k, n), **f) for _ in range(3))   # :73-74 (+ the loop's scalar temporaries)
        self.XHt, self.WHHt, self.tW = (np.empty((p, k), **f) for _ in range(3))   # :75-76
        self.preW, self.preH = np.empty((p, k), **f), np.empty((k, n), **f)        # nmf_skeleton!: common.jl:52-53
        self.sW = np.empty((p, k), **f)                           # stop_condition's running sums (scalar registers in Julia)
        self.sH = np.empty((k, n), **f)

    def update(self, X, W, H, phases=None):
        import time as _t
        T, o = self.T, self.o
        lw, lh, d = T(o.lambda_w), T(o.lambda_h), T(o.delta)

        def ph(name, f):
            if phases is None:
                f()
                return
            t0 = _t.perf_counter()
            f()
            phases[name] = phases.get(name, 0.0) + (_t.perf_counter() - t0)

        def upd(F, num, den, tmp, lam):
            np.subtract(num, lam, out=tmp)
            np.maximum(T(0), tmp, out=tmp)
            np.add(den, d, out=den)
            np.divide(tmp, den, out=tmp)
            np.multiply(F, tmp, out=F)

        if o.update_H:
            ph("mul! WtX = W'X      :98", lambda: np.matmul(W.T, X, out=self.WtX))
            ph("mul! WtWH = W'WH    :99", lambda: np.matmul(W.T, self.WH, out=self.WtWH))
            ph("H loop              :101-103", lambda: upd(H, self.WtX, self.WtWH, self.tH, lh))
            ph("mul! WH = W H       :104", lambda: np.matmul(W, H, out=self.WH))
        ph("mul! XHt = X H'     :109", lambda: np.matmul(X, H.T, out=self.XHt))
        ph("mul! WHHt = WH H'   :110", lambda: np.matmul(self.WH, H.T, out=self.WHHt))
        ph("W loop              :112-114", lambda: upd(W, self.XHt, self.WHHt, self.tW, lw))
        ph("mul! WH = W H       :115", lambda: np.matmul(W, H, out=self.WH))

    def stop_condition(self, W, H, tol, sequential=True, phases=None):
        """stop_condition(W, preW, H, preH, tol) (src/common.jl:92-111) on the state's preW / preH, no allocations.
        sequential=True: the strict left-to-right T-precision sums of the Julia loops (np.cumsum), bit-compatible with
        `stop_condition` above.  sequential=False: the same sums by np.sum (pairwise) -- one pass instead of a serial recurrence,
        which is what a compiled scalar loop costs; used for TIMING only (bench.py's cpu_baseline), where the emulated recurrence
        would bill the CPU for the emulation."""
        import time as _t
        T = self.T
        tol = T(tol)

        def sums(a, b, axis, tmp):
            np.subtract(a, b, out=tmp)
            np.multiply(tmp, tmp, out=tmp)
            if sequential:
                np.cumsum(tmp, axis=axis, dtype=T, out=tmp)
                dev = (tmp[-1, :] if axis == 0 else tmp[:, -1]).copy()
            else:
                dev = tmp.sum(axis=axis, dtype=T)
            np.add(a, b, out=tmp)
            np.multiply(tmp, tmp, out=tmp)
            if sequential:
                np.cumsum(tmp, axis=axis, dtype=T, out=tmp)
                sm = (tmp[-1, :] if axis == 0 else tmp[:, -1]).copy()
            else:
                sm = tmp.sum(axis=axis, dtype=T)
            return dev, sm

        t0 = _t.perf_counter()
        dw, sw = sums(W, self.preW, 0, self.sW)
        t1 = _t.perf_counter()
        dh, sh = sums(H, self.preH, 1, self.sH)
        t2 = _t.perf_counter()
        if phases is not None:
            phases["stop_condition W sums  common.jl:95-99"] = phases.get("stop_condition W sums  common.jl:95-99", 0.0) + (t1 - t0)
            phases["stop_condition H sums  common.jl:100-104"] = phases.get("stop_condition H sums  common.jl:100-104", 0.0) + (t2 - t1)
        bad = (np.sqrt(dw) > tol * np.sqrt(sw)) | (np.sqrt(dh) > tol * np.sqrt(sh))
        return not bool(np.any(bad))


class _MultDiv:
    """src/multupd.jl:121-193."""

    def __init__(self, T, o, X, W, H):
        self.T, self.o = T, o
        self.WH = W @ H

    def objv(self, X, W, H):
        return float(self.T(gkldiv(X, self.WH)))                  # :148

    def _ratio(self, X, d):
        """Q = X ./ (WH + delta) (:172-174, :184-186).  Large inputs: into ONE buffer kept across calls, like MultUpdDiv_State's Q
        (src/multupd.jl:128-147) -- same element-wise arithmetic, no fresh 1 GiB temporaries per pass."""
        if X.size <= _BLOCKED_ABOVE:
            return X / (self.WH + d)
        if getattr(self, "_Q", None) is None or self._Q.shape != X.shape:
            self._Q = np.empty_like(X)
        Q, WH = self._Q, self.WH

        def blk(x, q):           # q is a view into Q; its columns [j0, j1) are recovered from the view's offset
            j0 = (q.__array_interface__["data"][0] - Q.__array_interface__["data"][0]) // Q.strides[1]
            w = WH[:, j0:j0 + q.shape[1]]
            np.add(w, d, out=q)
            np.divide(x, q, out=q)
            return 0

        _pmap(blk, _col_blocks(X, Q))
        return Q

    def update(self, X, W, H):
        T, o = self.T, self.o
        lw, lh, d = T(o.lambda_w), T(o.lambda_h), T(o.delta)
        if o.update_H:
            Q = self._ratio(X, d)                                 # :172-174
            WtQ = W.T @ Q                                         # :175
            sW = np.cumsum(W, axis=0, dtype=W.dtype)[-1, :]       # :176 sum! in T
            H *= WtQ / (sW + lh)[:, None]                         # :177-179
            self.WH = W @ H                                       # :180
        Q = self._ratio(X, d)                                     # :184-186
        QHt = Q @ H.T                                             # :187
        sH = np.cumsum(H, axis=1, dtype=H.dtype)[:, -1]           # :188
        W *= QHt / (sH + lw)[None, :]                             # :189-191
        self.WH = W @ H                                           # :192


class _ProjALS:
    """src/projals.jl:42-107."""

    def __init__(self, T, o, X, W, H):
        self.T, self.o = T, o
        self.WH = W @ H

    def objv(self, X, W, H):
        T, o = self.T, self.o
        r = 0.5 * sqL2dist(X, self.WH)                            # :66  (T(0.5)*Float64 -> Float64)
        if o.lambda_w > 0:                                        # :67-69  (term in T, sum in Float64)
            r += float((T(0.5) * T(o.lambda_w)) * T(np.linalg.norm(W.ravel())) ** 2)
        if o.lambda_h > 0:                                        # :70-72
            r += float((T(0.5) * T(o.lambda_h)) * T(np.linalg.norm(H.ravel())) ** 2)
        return float(T(r))                                        # Result{T} conversion (common.jl:33)

    def update(self, X, W, H):
        T, o = self.T, self.o
        if o.update_H:
            WtW = adddiag(np.asfortranarray(W.T @ W), o.lambda_h)  # :92
            H[...] = pdsolve(WtW, np.asfortranarray(W.T @ X))     # :93-94
            projectnn(H)                                          # :95
        HHt = adddiag(np.asfortranarray(H @ H.T), o.lambda_w)     # :100
        XHt = X @ H.T                                             # :101
        W[...] = pdrsolve(XHt, HHt)                               # :102
        projectnn(W)                                              # :103
        self.WH = W @ H                                           # :106


def projgradnorm(g, x):
    """src/alspgrad.jl:9-19; sequential accumulation in T."""
    m = (g < 0) | (x > 0)
    v = np.where(m, g * g, g.dtype.type(0)).ravel(order="F")
    return g.dtype.type(np.sqrt(np.cumsum(v, dtype=g.dtype)[-1]))


def _dot(a, b):
    """BLAS.dot on the flattened (column-major) arrays."""
    return a.dtype.type(np.dot(a.ravel(order="F"), b.ravel(order="F")))


def _pgrad_subsolve(Z, Gram, B, left, maxiter, traceiter, tolg, beta, sigma, T, cnt):
    """_alspgrad_updateh! (alspgrad.jl:86-191, left=True: G = Gram*Z - B) and
    _alspgrad_updatew! (:242-347, left=False: G = Z*Gram - B).  Z is updated in
    place; returns the number of executed iterations t."""
    t = 0
    converged = False
    decr_alpha = True
    alpha = T(1)
    beta, sigma = T(beta), T(sigma)
    epsT = T(eps(T))
    Zp = None
    while (not converged) and t < maxiter:
        t += 1
        G = (Gram @ Z if left else Z @ Gram) - B                  # :124-127 / :280-283
        pgnrm = projgradnorm(G, Z)                                # :130
        if pgnrm < T(tolg):
            converged = True
        it = 0
        if not converged:
            while it < traceiter:
                it += 1
                cnt["backtracks"] += 1
                if not np.isfinite(alpha):
                    raise FloatingPointError("alpha is not finite")   # :140
                Zn = np.maximum(Z - alpha * G, T(0))              # :143-147
                D = Zn - Z
                dv1 = _dot(G, D)                                  # :150
                GD = Gram @ D if left else D @ Gram               # :151
                dv2 = _dot(GD, D)                                 # :152
                suff_decr = bool(((T(1) - sigma) * dv1 + T(0.5) * dv2) < 0)   # :155
                if it == 1:                                       # :157-160
                    decr_alpha = not suff_decr
                    Zp = Z.copy()
                if decr_alpha:
                    if suff_decr:                                 # :163-165
                        Z[...] = Zn
                        break
                    alpha = T(alpha * beta)                       # :167
                else:
                    diff = (Zp - Zn).ravel(order="F")
                    approx = bool(T(np.linalg.norm(diff)) <= epsT)  # isapprox atol=eps(T), rtol=0
                    if (not suff_decr) or approx:                 # :170-172
                        Z[...] = Zp
                        break
                    alpha = T(alpha / beta)                       # :174
                    Zp = Zn.copy()                                # :175
        cnt["inner"] += 1
    return t


class _ALSPGrad:
    """src/alspgrad.jl:352-425."""

    def __init__(self, T, o, X, W, H):
        self.T, self.o = T, o
        self.WH = W @ H
        self.tolg = T(o.tolg)                                     # fresh ALSPGradUpd per solve! (:381-383)
        self.cnt = {"inner": 0, "backtracks": 0}

    def objv(self, X, W, H):
        return float(self.T(0.5 * sqL2dist(X, self.WH)))          # :398

    def update(self, X, W, H):
        T, o = self.T, self.o
        if o.update_H:
            WtW = W.T @ W                                         # set_w! :63-67
            WtX = W.T @ X
            itH = _pgrad_subsolve(H, WtW, WtX, True, o.maxsubiter, o.traceiter,
                                  self.tolg, o.beta, o.sigma, T, self.cnt)   # :406-407
            if itH == 1:
                self.tolg = T(np.float64(self.tolg) * 0.1)        # :409-411 (Float64 literal)
        HHt = H @ H.T                                             # set_h! :218-222
        XHt = X @ H.T
        itW = _pgrad_subsolve(W, HHt, XHt, False, o.maxsubiter, o.traceiter,
                              self.tolg, o.beta, o.sigma, T, self.cnt)       # :416-417
        if itW == 1:
            self.tolg = T(np.float64(self.tolg) * 0.1)            # :419-421
        self.WH = W @ H                                           # :424


# ----------------------------------------------------------------------------
# CoordinateDescent (src/coorddesc.jl) and GreedyCD (src/greedycd.jl) -- SURVEY.md section 8f rank 2
# ----------------------------------------------------------------------------

def _update_coord_descent(X, W, H, l1_reg, l2_reg, permutation=None):
    """_update_coord_descent! (coorddesc.jl:107-158): updates W in place, returns the violation.  permutation = None is
    shuffle = false (:132-133); otherwise the component order of this call (:130-131, given instead of drawn).
    `H` is k x n (any strides).  The reference's t-outer / i-inner loops are kept; the inner loop over samples i is
    vectorised (rows do not interact) and the sum over r runs left to right in T exactly like :143-145."""
    T = X.dtype.type
    Ht = H.T
    HHt = H @ Ht                                                   # :111
    XHt = X @ Ht                                                   # :117
    k = H.shape[0]
    if l2_reg > 0:
        HHt[np.diag_indices(k)] += T(l2_reg)                       # :118-120
    if l1_reg > 0:
        XHt = XHt - T(l1_reg)                                      # :121-123
    violation = T(0)
    for t in (range(k) if permutation is None else permutation):   # :130-135
        grad = -XHt[:, t]                                          # :141
        for r in range(k):                                         # :143-145
            grad = grad + HHt[t, r] * W[:, r]
        pg = np.where(W[:, t] == 0, np.minimum(T(0), grad), grad)  # :148
        for v in np.abs(pg):                                       # :149  (sequential accumulation over i, in T)
            violation = T(violation + v)
        hess = HHt[t, t]                                           # :152
        if hess != 0:
            W[:, t] = np.maximum(W[:, t] - grad / hess, T(0))      # :153-155
    return violation


class _CoordDesc:
    """src/coorddesc.jl:84-181."""

    def __init__(self, T, o, X, W, H):
        self.T, self.o = T, o
        self.violation = T(0)
        self.calls = 0                                             # _update_coord_descent! calls so far (index of the next permutation)

    def objv(self, X, W, H):
        return float(self.T(0.5 * sqL2dist(X, W @ H)))             # :101-104

    def _perm(self, side):
        c = self.calls
        self.calls += 1
        return None if self.o.perm_source is None else [int(v) for v in self.o.perm_source(c - c % 2 + side)]

    def update(self, X, W, H):
        T, o = self.T, self.o
        v = _update_coord_descent(X, W, H, o.l1_w, o.l2_w, self._perm(0))        # :166
        if o.update_H:
            Ht = H.T                                                                # a view: the update lands in H
            v = T(v + _update_coord_descent(X.T, Ht, W.T, o.l1_h, o.l2_h, self._perm(1)))   # :169-174
        else:
            self.calls += 1                                                         # keep call_index = 2*(t-1) + side
        self.violation = v


def _greedy_sd(w, g, prr, T):
    """S and D of greedycd.jl:122-123 / :151-152 (elementwise, operation by operation in T)."""
    s = np.maximum(T(0), w - g / (T(eps(T)) + prr)) - w
    d = -g * s - (T(0.5) * prr) * (s * s)
    return s, d


def _update_greedycd(X, W, Ht, lam, counters):
    """_update_GreedyCD! (greedycd.jl:91-163): W (samples x k) updated in place from Ht (other-side samples x k)."""
    T = X.dtype.type
    ns, k = W.shape
    P = Ht.T @ Ht                                                  # :108
    Z = X @ Ht                                                     # :109
    G = W @ P - Z                                                  # :110-111
    if lam > 0:
        G = G + T(lam)                                             # :112-114
    prr = np.diag(P).copy()
    S, D = _greedy_sd(W, G, prr[None, :], T)                       # :117-122
    q = np.argmax(D, axis=1)                                       # :125-129 (first maximum, like Julia's argmax)
    p_init = max(T(-1.0), D[np.arange(ns), q].max()) if ns else T(-1.0)
    Wnew = np.zeros_like(W)                                        # :131
    nu = T(0.001)
    thresh = T(nu * p_init)
    steps = 0
    for i in range(ns):                                            # :134
        qi = int(q[i])
        w, g, s, d = W[i].copy(), G[i].copy(), S[i].copy(), D[i].copy()
        for _ in range(k * k):                                     # :136
            if d[qi] < thresh:                                     # :137-139
                break
            Wnew[i, qi] += s[qi]                                   # :141
            g = g + s[qi] * P[qi]                                  # :143-145
            s, d = _greedy_sd(w, g, prr, T)                        # :147-150
            qi = int(np.argmax(d))                                 # :152
            steps += 1
    counters["greedy_steps"] = counters.get("greedy_steps", 0) + steps
    W[...] = np.maximum(W + Wnew, T(0))                            # :156-157  copyto!(W, W + Wnew); projectnn!(W)


class _GreedyCD:
    """src/greedycd.jl:37-177."""

    def __init__(self, T, o, X, W, H):
        self.T, self.o = T, o
        self.cnt = {}

    def objv(self, X, W, H):
        T, o = self.T, self.o
        r = 0.5 * sqL2dist(X, W @ H)                               # :79-80
        if o.lambda_w > 0:
            r += float(T(T(o.lambda_w) * T(np.abs(W).sum(dtype=np.float64))))   # :81-83  lambda_w * norm(W, 1)
        if o.lambda_h > 0:
            r += float(T(T(o.lambda_h) * T(np.abs(H).sum(dtype=np.float64))))   # :84-86
        return float(T(r))

    def update(self, X, W, H):
        o = self.o
        _update_greedycd(X, W, H.T, o.lambda_w, self.cnt)          # :168
        if o.update_H:
            Ht = H.T                                               # view: in-place update of H
            _update_greedycd(X.T, Ht, W, o.lambda_h, self.cnt)     # :171-174


_UPDATERS = {MULTMSE: _MultMSE, MULTDIV: _MultDiv, PROJALS: _ProjALS, ALSPGRAD: _ALSPGrad, CD: _CoordDesc, GREEDYCD: _GreedyCD}


def alspgrad_updateh(X, W, H, maxiter=1000, traceiter=20, tolg=None, beta=0.2, sigma=0.01):
    """Public wrapper alspgrad_updateh! (alspgrad.jl:69-84); tolg default cbrt(eps(T))."""
    T = H.dtype.type
    tolg = T(np.cbrt(eps(T))) if tolg is None else T(tolg)
    cnt = {"inner": 0, "backtracks": 0}
    return _pgrad_subsolve(H, W.T @ W, W.T @ X, True, maxiter, traceiter, tolg, beta, sigma, T, cnt)


def alspgrad_updatew(X, W, H, maxiter=1000, traceiter=20, tolg=None, beta=0.2, sigma=0.01):
    """Public wrapper alspgrad_updatew! (alspgrad.jl:225-240)."""
    T = W.dtype.type
    tolg = T(np.cbrt(eps(T))) if tolg is None else T(tolg)
    cnt = {"inner": 0, "backtracks": 0}
    return _pgrad_subsolve(W, H @ H.T, X @ H.T, False, maxiter, traceiter, tolg, beta, sigma, T, cnt)


def nmf_checksize(X, W, H):
    """src/common.jl:5-16."""
    p, n = X.shape
    k = W.shape[1]
    if not (W.shape[0] == p and H.shape == (k, n)):
        raise ValueError("Dimensions of X, W, and H are inconsistent.")
    return p, n, k


def solve(alg, X, W, H, opts: Opts | None = None) -> Result:
    """NMF.solve!(alg, X, W, H) == nmf_skeleton! (src/common.jl:45-89).
    W and H are updated IN PLACE (they must be Fortran-ordered arrays of X's dtype)."""
    if isinstance(alg, str):
        alg = ALG_NAMES[alg]
    T = X.dtype.type
    assert W.dtype == X.dtype and H.dtype == X.dtype
    assert W.flags.f_contiguous and H.flags.f_contiguous
    nmf_checksize(X, W, H)
    o = resolve_opts(alg, T, opts or Opts())
    upd = _UPDATERS[alg](T, o, X, W, H)                           # prepare_state :51
    trace, relchange = [], []
    if o.track_objective:
        trace.append(upd.objv(X, W, H))                           # :56
        relchange.append(float("nan"))
    converged = False
    t = 0
    while (not converged) and t < o.maxiter:                      # :64
        t += 1
        preW = W.copy(order="F")                                  # :66-67
        preH = H.copy(order="F")
        upd.update(X, W, H)                                       # :70
        if o.track_objective:
            converged, dev = stop_condition_dev(W, preW, H, preH, o.tol)   # :73
            relchange.append(dev)
            trace.append(upd.objv(X, W, H))                       # :79
        else:
            converged = stop_condition(W, preW, H, preH, o.tol)   # :73
    objv = trace[-1] if (o.track_objective and trace) else upd.objv(X, W, H)   # :85-87
    return Result(W, H, t, converged, objv, trace, dict(getattr(upd, "cnt", {})), relchange)


# ----------------------------------------------------------------------------
# NNDSVD from a given truncated SVD (src/initialization.jl:26-137) -- SURVEY.md section 8f rank 3, the part behind
# `U, s, V = ...`.  The default `rsvd(X, k)` (RandomizedLinAlg, un-vendored, Julia RNG) is NOT restated: PARITY UNPINNED
# for it; callers pass `initdata` (any SVD of X), as test/initialization.jl:45-49 and test/interf.jl:20 do.
# ----------------------------------------------------------------------------

def _posnegnorm(x):
    """posnegnorm (:103-115): sequential T-precision sums of squares of the positive / non-positive entries."""
    T = x.dtype.type
    sq = x * x
    pn = np.cumsum(np.where(x > 0, sq, T(0)), dtype=T)[-1] if x.size else T(0)
    nn = np.cumsum(np.where(x > 0, T(0), sq), dtype=T)[-1] if x.size else T(0)
    return T(np.sqrt(pn)), T(np.sqrt(nn))


def nndsvd(X, k, zeroh=False, variant="std", initdata=None, rand_vj=None):
    """nndsvd(X, k; zeroh, variant, initdata) (:74-101) with _nndsvd! (:26-72).  initdata = (U, s, V) with X ~ U diag(s) V'
    (V is n x k); rand_vj: the k uniforms `rand(T)` draws for variant :ar (the caller supplies the stream)."""
    T = X.dtype.type
    p, n = X.shape
    ivar = {"std": 0, "a": 1, "ar": 2}.get(variant)
    if ivar is None:
        raise ValueError("Invalid value for variant")
    if initdata is None:
        raise NotImplementedError("rsvd(X, k) is not restated (un-vendored RandomizedLinAlg + Julia RNG): pass initdata")
    U, s, V = (np.asarray(a)[..., :k].astype(T) for a in initdata)                    # :83, :31-33
    W = np.empty((p, k), dtype=T, order="F")
    Ht = np.zeros((n, k), dtype=T, order="F")
    mean = np.mean(X, dtype=np.float64)
    v0 = T(0) if ivar == 0 else (T(mean) if ivar == 1 else T(mean * 0.01))           # :41-42
    for j in range(k):                                                                # :44
        x, y = U[:, j], V[:, j]
        xp, xn = _posnegnorm(x)
        yp, yn = _posnegnorm(y)
        mp, mn = T(xp * yp), T(xn * yn)                                               # :49-50
        vj = v0
        if ivar == 2:
            vj = T(vj * T(rand_vj[j]))                                                # :52-55
        if mp >= mn:                                                                  # :58-61 / :68-70
            ss = T(np.sqrt(T(s[j] * mp)))
            W[:, j] = np.where(x > 0, x * T(ss / xp), vj)
            if not zeroh:
                Ht[:, j] = np.where(y > 0, y * T(ss / yp), vj)
        else:                                                                         # :62-65 / :71-73
            ss = T(np.sqrt(T(s[j] * mn)))
            W[:, j] = np.where(x < 0, -(x * T(ss / xn)), vj)
            if not zeroh:
                Ht[:, j] = np.where(y < 0, -(y * T(ss / yn)), vj)
    H = np.zeros((k, n), dtype=T, order="F") if zeroh else np.asfortranarray(Ht.T)    # :89-99
    return W, H


# ----------------------------------------------------------------------------
# test problem (test/testproblems.jl:6-13)
# ----------------------------------------------------------------------------

# ----------------------------------------------------------------------------
# SPA -- successive projection algorithm for separable NMF (src/spa.jl:38-63), the last `init` / `alg` option of nnmf
# (src/interf.jl:50-51, 73-77).  H = nonneg_lsq(W, X, alg=:fnnls) comes from NonNegLeastSquares.jl, which is NOT vendored in the
# reference (Project.toml compat 0.4): it is Bro & de Jong's fast NNLS, an active-set method that returns THE minimiser of
# ||X[:, j] - W h||, h >= 0 (unique when W has full column rank), so any exact NNLS solver restates it up to rounding -- here
# SciPy's Lawson-Hanson `nnls`.  Pinned by test/spa.jl:11-32 (tests/test_oracle_kat.py).
# ----------------------------------------------------------------------------
def spa_anchors(X, k):
    """The anchor indices of spa() (src/spa.jl:40-55), 0-based."""
    T = X.dtype.type
    R = X / X.sum(axis=0, keepdims=True)                               # :41  columns of R sum to one
    ai = []
    for _ in range(k):
        a = int(np.argmax((R ** 2).sum(axis=0)))                       # :50  first index on ties, like Julia's argmax
        ai.append(a)
        pc = R[:, a].copy()                                            # :53
        R = R - np.outer(pc, pc @ R) / T(pc @ pc)                      # :54  p*(p'R) ./ (p'p)
    return ai


def spa(X, k):
    """W, H = spa(X, k) (src/spa.jl:38-63)."""
    from scipy.optimize import nnls
    T = X.dtype.type
    ai = spa_anchors(X, k)
    W = np.asfortranarray(X[:, ai])                                    # :58
    W64 = W.astype(np.float64)
    H = np.zeros((k, X.shape[1]), dtype=T, order="F")
    for j in range(X.shape[1]):                                        # :61  nonneg_lsq(W, X, alg=:fnnls), column by column
        H[:, j] = nnls(W64, X[:, j].astype(np.float64))[0].astype(T)
    projectnn(H)                                                       # :62
    return W, H, ai


def spa_solve(X, W, H, obj="mse"):
    """solve!(::SPA, X, W, H) (src/spa.jl:66-75): no iterations, only the objective."""
    T = X.dtype.type
    objv = 0.5 * sqL2dist(X, W @ H) if obj == "mse" else gkldiv(X, W @ H)
    return Result(W, H, 0, True, float(T(objv)))


def laurberg6x3(alpha, T=np.float64):
    a = T(alpha)
    H = np.array([[a, 1, 1, a, 0, 0],
                  [1, a, 0, 0, a, 1],
                  [0, 0, a, 1, 1, a]], dtype=T, order="F")
    W = np.asfortranarray(H.T.copy())
    X = np.asfortranarray(W @ H)
    return X, W, H
