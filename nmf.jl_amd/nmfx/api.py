"""Host-side mirror of the NMF.jl operator interface for the accelerated path.

Same names, argument meaning and error behaviour as the reference so the parity
tests read like the reference's own tests:

    nnmf(X, k; init, alg, maxiter, tol, replicates, W0, H0, update_H, verbose)   src/interf.jl:3-83
    solve(alg, X, W, H) -> Result        NMF.solve!             src/multupd.jl:45, projals.jl:37, alspgrad.jl:381
    MultUpdate / ProjectedALS / ALSPGrad option structs         src/multupd.jl:9-42, projals.jl:18-34, alspgrad.jl:352-373
    Result(W, H, niters, converged, objvalue)                   src/common.jl:21-38
    alspgrad_updateh / alspgrad_updatew                         src/alspgrad.jl:69-84, 225-240

In production the host stays in Julia (nmf.jl_amd/julia/NMFX.jl does the same over
`ccall`); this Python twin exists because the build image has no Julia.  All numeric
work happens in libnmfx.so on the GPU -- this module only validates, marshals and
maps status codes to the reference's exception types.
"""
from __future__ import annotations

import ctypes as C
import math
import warnings
from dataclasses import dataclass

import numpy as np

from . import _lib as L


# ---- the reference's exception vocabulary -------------------------------------------------
class ArgumentError(ValueError):
    """Julia ArgumentError (src/interf.jl:15-33, src/multupd.jl:27-31)."""


class DimensionMismatch(ValueError):
    """Julia DimensionMismatch (src/common.jl:12,30)."""


class PosDefException(np.linalg.LinAlgError):
    """LinearAlgebra.PosDefException raised by potrf! (src/utils.jl:68,78)."""


class NMFXError(RuntimeError):
    """HIP / RCCL / state errors of the library itself."""


def _raise(status: int, msg: str):
    if status == L.ERR_BAD_ARG:
        raise ArgumentError(msg)
    if status == L.ERR_DIM_MISMATCH:
        raise DimensionMismatch(msg)
    if status == L.ERR_NOT_POSDEF:
        raise PosDefException(msg)
    if status == L.ERR_ALPHA_NONFINITE:
        raise RuntimeError("α is not finite")          # error("α is not finite"), src/alspgrad.jl:140
    raise NMFXError(f"nmfx status {status}: {msg}")


def _eps(T):
    return float(np.finfo(T).eps)


# ---- option structs --------------------------------------------------------------------
class MultUpdate:
    """MultUpdate{T}(; obj, maxiter, verbose, tol, update_H, lambda_w, lambda_h)  (src/multupd.jl:9-42)."""

    def __init__(self, T, obj="mse", maxiter=100, verbose=False, tol=None, update_H=True,
                 lambda_w=0.0, lambda_h=0.0, lambda_=None):
        T = np.dtype(T).type
        tol = float(T(np.cbrt(_eps(T)))) if tol is None else tol
        if obj not in ("mse", "div"):
            raise ArgumentError("Invalid value for obj.")
        if not maxiter > 1:
            raise ArgumentError("maxiter must be greater than 1.")
        if not tol > 0:
            raise ArgumentError("tol must be positive.")
        if not lambda_w >= 0:
            raise ArgumentError("lambda_w must be non-negative.")
        if not lambda_h >= 0:
            raise ArgumentError("lambda_h must be non-negative.")
        if lambda_ is not None and lambda_ >= 0:
            warnings.warn("lambda is deprecated, use lambda_w and lambda_h instead.")
            lambda_w = lambda_ if lambda_w == 0 else lambda_w
            lambda_h = lambda_ if lambda_h == 0 else lambda_h
        if obj == "div":
            lambda_w = max(lambda_w, math.sqrt(_eps(T)))
            lambda_h = max(lambda_h, math.sqrt(_eps(T)))
        self.T, self.obj, self.maxiter, self.verbose = T, obj, int(maxiter), bool(verbose)
        self.tol, self.update_H = float(T(tol)), bool(update_H)
        self.lambda_w, self.lambda_h = float(T(lambda_w)), float(T(lambda_h))

    def _alg(self):
        return L.ALG_MULTMSE if self.obj == "mse" else L.ALG_MULTDIV

    def _opts(self):
        return dict(maxiter=self.maxiter, tol=self.tol, update_H=self.update_H, lambda_w=self.lambda_w,
                    lambda_h=self.lambda_h, delta=float(self.T(math.sqrt(_eps(self.T)))))   # src/multupd.jl:48,50


class ProjectedALS:
    """ProjectedALS{T}(; maxiter, verbose, tol, update_H, lambda_w, lambda_h)  (src/projals.jl:18-34); no validation."""

    def __init__(self, T, maxiter=100, verbose=False, tol=None, update_H=True, lambda_w=None, lambda_h=None):
        T = np.dtype(T).type
        d = float(T(np.cbrt(_eps(T))))
        self.T, self.maxiter, self.verbose = T, int(maxiter), bool(verbose)
        self.tol = float(T(d if tol is None else tol))
        self.update_H = bool(update_H)
        self.lambda_w = float(T(d if lambda_w is None else lambda_w))
        self.lambda_h = float(T(d if lambda_h is None else lambda_h))

    def _alg(self):
        return L.ALG_PROJALS

    def _opts(self):
        return dict(maxiter=self.maxiter, tol=self.tol, update_H=self.update_H, lambda_w=self.lambda_w,
                    lambda_h=self.lambda_h)


class ALSPGrad:
    """ALSPGrad{T}(; maxiter, maxsubiter, tol, tolg, update_H, verbose)  (src/alspgrad.jl:352-373).
    gradient (no reference counterpart): "exact" = G = Gram*Z - B by a full product every inner iteration, the reference's form
    (src/alspgrad.jl:124-127); an integer n > 1 = running gradient with a full product every n-th inner iteration; None = the
    library default (include/nmfx.h: nmfx_opts.pg_refresh)."""

    def __init__(self, T, maxiter=100, maxsubiter=200, tol=None, tolg=None, update_H=True, verbose=False, gradient=None):
        T = np.dtype(T).type
        self.T, self.maxiter, self.maxsubiter = T, int(maxiter), int(maxsubiter)
        self.tol = float(T(np.cbrt(_eps(T)) if tol is None else tol))
        self.tolg = float(T(_eps(T) ** 0.25 if tolg is None else tolg))
        self.update_H, self.verbose = bool(update_H), bool(verbose)
        if gradient is not None and gradient != "exact" and not (isinstance(gradient, int) and gradient >= 1):
            raise ValueError("gradient must be None, 'exact' or an integer >= 1")
        self.pg_refresh = 0 if gradient is None else (1 if gradient == "exact" else int(gradient))

    def _alg(self):
        return L.ALG_ALSPGRAD

    def _opts(self):
        return dict(maxiter=self.maxiter, tol=self.tol, update_H=self.update_H, maxsubiter=self.maxsubiter,
                    tolg=self.tolg, pg_refresh=self.pg_refresh)


class CoordinateDescent:
    """CoordinateDescent{T}(; maxiter, verbose, tol, update_H, α, regularization, l₁ratio, shuffle)  (src/coorddesc.jl:23-51);
    the resolved l1/l2 pairs are those of CoordinateDescentUpd (src/coorddesc.jl:62-82).  shuffle=True sweeps the components
    in a fresh random order per side and iteration (src/coorddesc.jl:130-134); Julia's randperm stream cannot be reproduced,
    the orders come from the documented Philox generator of include/nmfx.h, keyed by `shuffle_seed` (non-zero)."""

    def __init__(self, T, maxiter=100, verbose=False, tol=None, update_H=True, alpha=0.0, regularization="both",
                 l1ratio=0.0, shuffle=False, shuffle_seed=1):
        T = np.dtype(T).type
        if shuffle and (int(shuffle_seed) == 0 or not -2**31 <= int(shuffle_seed) < 2**31):
            raise ArgumentError("shuffle_seed must be a non-zero 32-bit integer")
        self.shuffle, self.shuffle_seed = bool(shuffle), int(shuffle_seed)
        if regularization not in ("both", "components", "transformation", "none"):
            raise ArgumentError("Invalid value for regularization.")
        self.T, self.maxiter, self.verbose = T, int(maxiter), bool(verbose)
        self.tol = float(T(np.cbrt(_eps(T)) if tol is None else tol))
        self.update_H = bool(update_H)
        a, l1r = T(alpha), T(l1ratio)
        aH = a if regularization in ("both", "components") else T(0)
        aW = a if regularization in ("both", "transformation") else T(0)
        self.l1_w, self.l2_w = float(T(aW * l1r)), float(T(aW * (T(1) - l1r)))
        self.l1_h, self.l2_h = float(T(aH * l1r)), float(T(aH * (T(1) - l1r)))

    def _alg(self):
        return L.ALG_CD

    def _opts(self):
        return dict(maxiter=self.maxiter, tol=self.tol, update_H=self.update_H, l1_w=self.l1_w, l2_w=self.l2_w,
                    l1_h=self.l1_h, l2_h=self.l2_h, cd_shuffle=self.shuffle_seed if self.shuffle else 0)


class GreedyCD:
    """GreedyCD{T}(; maxiter, verbose, tol, update_H, lambda_w, lambda_h)  (src/greedycd.jl:10-32)."""

    def __init__(self, T, maxiter=100, verbose=False, tol=None, update_H=True, lambda_w=0.0, lambda_h=0.0):
        T = np.dtype(T).type
        tol = float(T(np.cbrt(_eps(T)))) if tol is None else tol
        if not maxiter > 1:
            raise ArgumentError("maxiter must be greater than 1.")
        if not tol > 0:
            raise ArgumentError("tol must be positive.")
        if not lambda_w >= 0:
            raise ArgumentError("lambda_w must be non-negative.")
        if not lambda_h >= 0:
            raise ArgumentError("lambda_h must be non-negative.")
        self.T, self.maxiter, self.verbose = T, int(maxiter), bool(verbose)
        self.tol, self.update_H = float(T(tol)), bool(update_H)
        self.lambda_w, self.lambda_h = float(T(lambda_w)), float(T(lambda_h))

    def _alg(self):
        return L.ALG_GREEDYCD

    def _opts(self):
        return dict(maxiter=self.maxiter, tol=self.tol, update_H=self.update_H, lambda_w=self.lambda_w, lambda_h=self.lambda_h)


@dataclass
class SPA:
    """NMF.SPA{T}(obj = :mse) (src/spa.jl:26-34): no iterations, solve! only evaluates the objective of the given W, H."""
    T: type
    obj: str = "mse"

    def __post_init__(self):
        self.T = np.dtype(self.T).type
        if self.obj not in ("mse", "div"):
            raise ArgumentError("Invalid value for obj.")

    def _alg(self):
        return L.ALG_MULTMSE if self.obj == "mse" else L.ALG_MULTDIV

    def _opts(self):
        return dict(maxiter=0)


@dataclass
class Result:
    """NMF.Result{T} (src/common.jl:21-38).  W and H are the caller's arrays, updated in place."""
    W: np.ndarray
    H: np.ndarray
    niters: int
    converged: bool
    objvalue: float
    trace: np.ndarray | None = None
    info: dict | None = None

    def __post_init__(self):
        if self.W.shape[1] != self.H.shape[0]:
            raise DimensionMismatch("Inner dimensions of W and H mismatch.")

    def __eq__(self, o):
        return (np.array_equal(self.W, o.W) and np.array_equal(self.H, o.H) and self.niters == o.niters
                and self.converged == o.converged and self.objvalue == o.objvalue)

    def __hash__(self):
        return hash((self.W.tobytes(), self.H.tobytes(), self.niters, self.converged, self.objvalue))


def make_opts(T, maxiter=100, tol=None, update_H=True, lambda_w=0.0, lambda_h=0.0, delta=None, maxsubiter=200,
              traceiter=20, tolg=None, beta=0.2, sigma=0.01, track_objective=False, check_every=0,
              l1_w=0.0, l2_w=0.0, l1_h=0.0, l2_h=0.0, precision="fp32", cd_shuffle=0, pg_refresh=0, h_solve="auto", exact_stop=False) -> L.Opts:
    T = np.dtype(T).type
    return L.Opts(int(maxiter), int(bool(update_H)), int(bool(track_objective)), int(maxsubiter), int(traceiter),
                  int(check_every),
                  float(T(np.cbrt(_eps(T)) if tol is None else tol)), float(lambda_w), float(lambda_h),
                  float(T(math.sqrt(_eps(T))) if delta is None else delta),
                  float(T(_eps(T) ** 0.25) if tolg is None else tolg), float(T(beta)), float(T(sigma)),
                  float(l1_w), float(l2_w), float(l1_h), float(l2_h),
                  {"fp32": L.PREC_FP32, "bf16x3": L.PREC_BF16X3}[precision], int(cd_shuffle), int(pg_refresh),
                  {"auto": 0, "product": 1, "potrs": 2}[h_solve], int(bool(exact_stop)), 0)


def nmf_checksize(X, W, H):
    """src/common.jl:5-16."""
    p, n = X.shape
    k = W.shape[1]
    if not (W.shape[0] == p and H.shape == (k, n)):
        raise DimensionMismatch("Dimensions of X, W, and H are inconsistent.")
    return p, n, k


class Context:
    """Owns one nmfx_ctx: the device-resident X (uploaded once, src/interf.jl:85-101 re-uses it across
    replicates) and all solver temporaries (the reference's prepare_state objects)."""

    def __init__(self, dtype, p, n, k, device=0):
        self.lib = L.load()
        self.T = np.dtype(dtype).type
        if self.T not in (np.float32, np.float64):
            raise ArgumentError("element type must be Float32 or Float64")
        self.p, self.n, self.k = int(p), int(n), int(k)
        h = C.c_void_p()
        st = self.lib.nmfx_create(C.byref(h), L.F32 if self.T == np.float32 else L.F64, self.p, self.n, self.k, device)
        if st != L.OK:
            _raise(st, self.lib.nmfx_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.nmfx_destroy(self.h)
            self.h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, st):
        if st != L.OK:
            _raise(st, self.lib.nmfx_last_error(self.h).decode())

    def set_X(self, X):
        X = np.asfortranarray(X, dtype=self.T)
        assert X.shape == (self.p, self.n)
        self._ck(self.lib.nmfx_set_X(self.h, X.ctypes.data, X.shape[0]))

    def set_X_device(self, ptr, ldx):
        self._ck(self.lib.nmfx_set_X_device(self.h, C.c_void_p(ptr), ldx))

    def set_factors(self, W, H):
        assert W.flags.f_contiguous and H.flags.f_contiguous and W.dtype == self.T and H.dtype == self.T
        self._ck(self.lib.nmfx_set_factors(self.h, W.ctypes.data, H.ctypes.data))

    def get_factors(self, W=None, H=None):
        self._ck(self.lib.nmfx_get_factors(self.h, W.ctypes.data if W is not None else None,
                                           H.ctypes.data if H is not None else None))

    def iterate(self, alg, opts: L.Opts):
        res = L.CResult()
        trace = np.full(opts.maxiter + 1, np.nan) if opts.track_objective else None
        st = self.lib.nmfx_iterate(self.h, alg, C.byref(opts), C.byref(res), trace.ctypes.data if trace is not None else None)
        self._ck(st)
        return res, trace

    def solve(self, alg, opts: L.Opts, W, H):
        assert W.flags.f_contiguous and H.flags.f_contiguous and W.dtype == self.T and H.dtype == self.T
        if W.shape != (self.p, self.k) or H.shape != (self.k, self.n):
            raise DimensionMismatch("Dimensions of X, W, and H are inconsistent.")
        res = L.CResult()
        trace = np.full(opts.maxiter + 1, np.nan) if opts.track_objective else None
        st = self.lib.nmfx_solve(self.h, alg, C.byref(opts), W.ctypes.data, H.ctypes.data, C.byref(res),
                                 trace.ctypes.data if trace is not None else None)
        self._ck(st)
        return res, trace

    def subsolve(self, which, opts: L.Opts, W, H):
        res = L.CResult()
        self._ck(self.lib.nmfx_alspgrad_subsolve(self.h, which, C.byref(opts), W.ctypes.data, H.ctypes.data, C.byref(res)))
        return res

    def objective(self, alg, opts: L.Opts):
        out = C.c_double()
        self._ck(self.lib.nmfx_objective(self.h, alg, C.byref(opts), C.byref(out)))
        return out.value

    # ---- nnmf front end on the device (include/nmfx.h)
    def check_nonneg(self, which=0) -> bool:
        """all(t -> t >= zero(T), A) for A = X (0), W (1), H (2) (src/interf.jl:15, 28, 31)."""
        out = C.c_int32()
        self._ck(self.lib.nmfx_check_nonneg(self.h, which, C.byref(out)))
        return bool(out.value)

    def randinit(self, seed, normalize=False, zeroh=False, h_col_offset=0):
        """randinit(X, k; normalize, zeroh) (src/initialization.jl:4-17) into the resident W, H (Philox4x32-10)."""
        self._ck(self.lib.nmfx_randinit(self.h, seed, int(normalize), int(zeroh), h_col_offset))

    def solve_replicates(self, alg, opts: L.Opts, replicates, seed, zeroh, W, H, h_col_offset=0):
        """solve_replicates! (src/interf.jl:85-101); W, H are the replicate-1 start and receive the winner."""
        res, best = L.CResult(), C.c_int32()
        self._ck(self.lib.nmfx_solve_replicates(self.h, alg, C.byref(opts), replicates, seed, int(zeroh), h_col_offset,
                                                W.ctypes.data, H.ctypes.data, C.byref(res), C.byref(best)))
        return res, best.value

    def rsvd(self, seed=0, h_col_offset=0, download=True, power_iters=0):
        """rsvd(X, k) (src/initialization.jl:83) on the resident X: the device does the p*n*k products and the
        orthogonalisation, the k x k symmetric eigenproblem is solved here with LAPACK (like the reference's small svd).
        Returns (U, s, V) with V n x k when download, else None; the triple stays resident for nndsvd_init(None, ...)."""
        T, k = self.T, self.k
        Cm = np.empty((k, k), dtype=T, order="F")
        self._ck(self.lib.nmfx_rsvd_begin(self.h, seed, h_col_offset, power_iters, Cm.ctypes.data))
        ev, Ub = np.linalg.eigh(((Cm + Cm.T) * 0.5).astype(np.float64))
        order = np.argsort(ev)[::-1]
        s = np.sqrt(np.maximum(ev[order], 0.0)).astype(T)
        Ub = np.asfortranarray(Ub[:, order].astype(T))
        if not download:
            self._ck(self.lib.nmfx_rsvd_finish(self.h, Ub.ctypes.data, s.ctypes.data, None, None))
            return None
        U = np.empty((self.p, k), dtype=T, order="F")
        Vt = np.empty((k, self.n), dtype=T, order="F")
        self._ck(self.lib.nmfx_rsvd_finish(self.h, Ub.ctypes.data, s.ctypes.data, U.ctypes.data, Vt.ctypes.data))
        return U, s, np.asfortranarray(Vt.T)

    def nndsvd_init(self, U, s, V, variant="std", zeroh=False, seed=0, n_total=None):
        """_nndsvd! (src/initialization.jl:26-72) on the device from a given truncated SVD (U = s = V = None: the one
        left resident by rsvd()): fills the resident W, H."""
        T, k = self.T, self.k
        ivar = {"std": 0, "a": 1, "ar": 2}.get(variant)
        if ivar is None:
            raise ArgumentError("Invalid value for variant")
        if U is None:
            self._ck(self.lib.nmfx_nndsvd(self.h, None, None, None, ivar, int(zeroh), seed, self.n if n_total is None else n_total))
            return
        U = np.asfortranarray(np.asarray(U)[:, :k], dtype=T)
        V = np.asfortranarray(np.asarray(V)[:, :k], dtype=T)
        s = np.ascontiguousarray(np.asarray(s)[:k], dtype=T)
        if U.shape != (self.p, k) or V.shape != (self.n, k) or s.shape != (k,):
            raise DimensionMismatch("U must be p x k, s of length k, V n x k")
        ivar = {"std": 0, "a": 1, "ar": 2}.get(variant)
        if ivar is None:
            raise ArgumentError("Invalid value for variant")
        self._ck(self.lib.nmfx_nndsvd(self.h, U.ctypes.data, s.ctypes.data, V.ctypes.data, ivar, int(zeroh), seed,
                                      self.n if n_total is None else n_total))

    def iter_trace(self, count):
        """(elapsed seconds, (W & H).relchange) per iteration of the last tracked solve (src/common.jl:76-82)."""
        el, rc, m = np.zeros(count), np.full(count, np.nan), C.c_int32()
        self._ck(self.lib.nmfx_get_iter_trace(self.h, el.ctypes.data, rc.ctypes.data, count, C.byref(m)))
        return el[: m.value], rc[: m.value]

    def comm_init(self, uid: bytes, rank: int, nranks: int):
        buf = C.create_string_buffer(uid, L.UNIQUE_ID_BYTES)
        self._ck(self.lib.nmfx_comm_init(self.h, buf, rank, nranks))

    def spa_init(self, warm_sweeps=16):
        """spa(X, k) (src/spa.jl:38-63) on the resident X into the resident W, H; returns (anchors, unsolved): the anchor column
        indices (0-based, selection order) and the number of columns whose active-set solve did not reach its KKT test."""
        anchors = np.empty(self.k, dtype=np.int64)
        unsolved = C.c_int64()
        self._ck(self.lib.nmfx_spa_init(self.h, int(warm_sweeps), anchors.ctypes.data, C.byref(unsolved)))
        return anchors, unsolved.value

    def pdsolve(self, A, B, lambda_=0.0, project_nn=False):
        """inv(A + lambda I) * B on the device kernels of ProjectedALS (adddiag! + pdsolve! [+ projectnn!], src/utils.jl); A is
        k x k SPD, B is k x n (the context's k, n)."""
        T = self.T
        A = np.asfortranarray(A, dtype=T)
        B = np.asfortranarray(B, dtype=T)
        if A.shape != (self.k, self.k) or B.shape != (self.k, self.n):
            raise DimensionMismatch("A must be k x k and B k x n")
        X = np.empty_like(B, order="F")
        self._ck(self.lib.nmfx_pdsolve(self.h, A.ctypes.data, float(lambda_), B.ctypes.data, X.ctypes.data, int(project_nn)))
        return X

    def pdrsolve(self, A, B, lambda_=0.0, project_nn=False):
        """A * inv(B + lambda I) (adddiag! + pdrsolve! [+ projectnn!]); A is p x k, B is k x k SPD."""
        T = self.T
        A = np.asfortranarray(A, dtype=T)
        B = np.asfortranarray(B, dtype=T)
        if A.shape != (self.p, self.k) or B.shape != (self.k, self.k):
            raise DimensionMismatch("A must be p x k and B k x k")
        X = np.empty_like(A, order="F")
        self._ck(self.lib.nmfx_pdrsolve(self.h, A.ctypes.data, B.ctypes.data, float(lambda_), X.ctypes.data, int(project_nn)))
        return X

    def comm_init_local(self, group: "LocalGroup", rank: int):
        """Attach this context as `rank` of an in-process group (collective: every rank calls it from its own thread)."""
        self._ck(self.lib.nmfx_comm_init_local(self.h, group.h, rank))

    def comm_init_sim(self, rank: int, nranks: int):
        """Timing stand-in: rank `rank` of `nranks` without peers (results are not a factorisation; bench.py --sim-ranks)."""
        self._ck(self.lib.nmfx_comm_init_sim(self.h, rank, nranks))

    def comm_init_p2p(self, rank: int, nranks: int):
        """Rank `rank` of `nranks` with the peer windows as the ONLY transport (no RCCL): follow with comm_p2p_export /
        comm_p2p_attach.  Works with several processes on one device (where RCCL refuses duplicate GPUs)."""
        self._ck(self.lib.nmfx_comm_init_p2p(self.h, rank, nranks))

    def comm_p2p_export(self) -> bytes:
        """Allocate this rank's exchange window and return its 128-byte handle (the host ships the handles of all ranks, in rank
        order, to every rank -- like RCCL's unique id).  Needs a communicator (any transport); it becomes the fallback for
        collectives the windows cannot serve."""
        buf = C.create_string_buffer(L.P2P_HANDLE_BYTES)
        self._ck(self.lib.nmfx_comm_p2p_export(self.h, buf))
        return buf.raw

    def comm_p2p_attach(self, handles):
        """Map every rank's window (handles: the nranks exported handles in rank order) and switch the exchange to them.
        handles = None detaches: every collective goes to the wrapped transport again."""
        if handles is None:
            self._ck(self.lib.nmfx_comm_p2p_attach(self.h, None))
            return
        blob = b"".join(handles)
        buf = C.create_string_buffer(blob, len(blob))
        self._ck(self.lib.nmfx_comm_p2p_attach(self.h, buf))

    def comm_p2p_stats(self):
        """(collectives served by the peer windows, collectives handed to the wrapped transport) so far."""
        w, b = C.c_int64(), C.c_int64()
        self._ck(self.lib.nmfx_comm_p2p_stats(self.h, C.byref(w), C.byref(b)))
        return w.value, b.value

    def comm_set_mode(self, mode: str):
        """'row_sharded' (default: reduce-scatter / all-gather, each rank updates its rows of W), 'replicated_w' (one packed
        all-reduce, every rank applies the full W update), 'pipelined' (opt-in, MultUpdate-MSE), or 'replicas': every rank holds the
        FULL X and solve_replicates deals the replicates out over the ranks (src/interf.jl:85-101; include/nmfx.h)."""
        self._ck(self.lib.nmfx_comm_set_mode(self.h, {"row_sharded": L.COMM_ROW_SHARDED, "replicated_w": L.COMM_REPLICATED_W, "pipelined": L.COMM_PIPELINED,
                                                       "replicas": L.COMM_REPLICAS}[mode]))

    def profile_enable(self, mode=1):
        """0 off, 1 every launch (slow), 2 dominant GEMMs sampled 1-in-8 (bench roofline)."""
        self._ck(self.lib.nmfx_profile_enable(self.h, int(mode)))

    def set_final_objective(self, on=True):
        """False: iterate() leaves Result.objvalue NaN (ask objective() afterwards) -- for callers that time K iterations."""
        self._ck(self.lib.nmfx_set_final_objective(self.h, int(bool(on))))

    def profile_get(self):
        arr = (L.KernelStat * 64)()
        cnt = C.c_int()
        self._ck(self.lib.nmfx_profile_get(self.h, arr, 64, C.byref(cnt)))
        return [dict(name=arr[i].name.decode(), ms_total=arr[i].ms_total, launches=arr[i].launches,
                     flops=arr[i].flops, bytes=arr[i].bytes) for i in range(cnt.value)]


class LocalGroup:
    """nmfx_local_group: the in-process transport of the sharded path (several contexts, one host thread each)."""

    def __init__(self, nranks: int):
        self.lib = L.load()
        h = C.c_void_p()
        st = self.lib.nmfx_local_group_create(C.byref(h), nranks)
        if st != L.OK:
            _raise(st, "nmfx_local_group_create failed")
        self.h, self.nranks = h, nranks

    def close(self):
        """Releases the group; if contexts are still attached the library frees it when the last of them closes."""
        if getattr(self, "h", None):
            self.lib.nmfx_local_group_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def comm_unique_id() -> bytes:
    lib = L.load()
    buf = C.create_string_buffer(L.UNIQUE_ID_BYTES)
    st = lib.nmfx_comm_get_unique_id(buf)
    if st != L.OK:
        _raise(st, "ncclGetUniqueId failed")
    return buf.raw


def _result(T, W, H, res, trace):
    tr = None if trace is None else trace[: res.niters + 1]
    return Result(W, H, int(res.niters), bool(res.converged), float(T(res.objvalue)), tr,
                  dict(seconds_loop=res.seconds_loop, inner_iters=res.inner_iters, backtracks=res.backtracks,
                       final_tolg=res.final_tolg))


def solve(alg, X, W, H, ctx: Context | None = None, track_objective=False, check_every=0, precision="fp32") -> Result:
    """NMF.solve!(alg, X, W, H): W and H (Fortran-ordered, dtype of X) are updated in place."""
    p, n, k = nmf_checksize(X, W, H)
    T = alg.T
    if X.dtype != T or W.dtype != T or H.dtype != T:
        raise ArgumentError("X, W, H must have the algorithm's element type")
    own = ctx is None
    if own:
        ctx = Context(T, p, n, k)
        ctx.set_X(X)
    track_objective = track_objective or bool(getattr(alg, "verbose", False))   # verbose = true evaluates every iteration
    try:
        if isinstance(alg, SPA):                                                 # src/spa.jl:66-75
            ctx.set_factors(W, H)
            return Result(W, H, 0, True, ctx.objective(alg._alg(), make_opts(T)), None, {})
        o = make_opts(T, track_objective=track_objective, check_every=check_every, precision=precision, **alg._opts())
        res, trace = ctx.solve(alg._alg(), o, W, H)
        out = _result(T, W, H, res, trace)
        if track_objective:
            out.info["elapsed"], out.info["relchange"] = ctx.iter_trace(int(res.niters) + 1)
        if getattr(alg, "verbose", False):
            print_verbose_table(out)
        return out
    finally:
        if own:
            ctx.close()


def print_verbose_table(r: Result, file=None):
    """The table nmf_skeleton! prints with verbose = true (src/common.jl:54-59, :76-82), same columns and formats."""
    el, rc = r.info["elapsed"], r.info["relchange"]
    print("%-5s    %-13s    %-13s    %-13s    %-13s" % ("Iter", "Elapsed time", "objv", "objv.change", "(W & H).relchange"), file=file)
    print("%5d    %13.6e    %13.6e" % (0, 0.0, r.trace[0]), file=file)
    for t in range(1, len(r.trace)):
        print("%5d    %13.6e    %13.6e    %13.6e    %13.6e" % (t, el[t], r.trace[t], r.trace[t] - r.trace[t - 1], rc[t]), file=file)


def alspgrad_updateh(X, W, H, maxiter=1000, traceiter=20, tolg=None, beta=0.2, sigma=0.01):
    """alspgrad_updateh!(X, W, H; ...) (src/alspgrad.jl:69-84); tolg default cbrt(eps(T)).  Returns iterations."""
    T = H.dtype.type
    p, n, k = nmf_checksize(X, W, H)
    with Context(T, p, n, k) as ctx:
        ctx.set_X(X)
        o = make_opts(T, maxsubiter=maxiter, traceiter=traceiter, tolg=float(T(np.cbrt(_eps(T)))) if tolg is None else tolg,
                      beta=beta, sigma=sigma)
        return int(ctx.subsolve(0, o, W, H).niters)


def alspgrad_updatew(X, W, H, maxiter=1000, traceiter=20, tolg=None, beta=0.2, sigma=0.01):
    """alspgrad_updatew!(X, W, H; ...) (src/alspgrad.jl:225-240)."""
    T = W.dtype.type
    p, n, k = nmf_checksize(X, W, H)
    with Context(T, p, n, k) as ctx:
        ctx.set_X(X)
        o = make_opts(T, maxsubiter=maxiter, traceiter=traceiter, tolg=float(T(np.cbrt(_eps(T)))) if tolg is None else tolg,
                      beta=beta, sigma=sigma)
        return int(ctx.subsolve(1, o, W, H).niters)


def randinit(X, k, normalize=False, zeroh=False, rng=None):
    """randinit (src/initialization.jl:4-17).  The Julia Xoshiro stream is not reproducible here;
    parity runs pass explicit W0/H0 (init=:custom)."""
    rng = np.random.default_rng() if rng is None else rng
    p, n = X.shape
    T = X.dtype.type
    W = np.asfortranarray(rng.random((p, k)).astype(T))
    if normalize:
        W /= W.sum(axis=0, keepdims=True)                 # normalize1_cols! (src/utils.jl:28-32)
    H = np.zeros((k, n), dtype=T, order="F") if zeroh else np.asfortranarray(rng.random((k, n)).astype(T))
    return W, H


_ALGS = ("greedycd", "cd", "multmse", "multdiv", "projals", "alspgrad")


def truncated_svd(X, k):
    """(U, s, V) with X ~ U diag(s) V', k leading triplets: the host-side stand-in for the reference's rsvd(X, k)
    (RandomizedLinAlg, src/initialization.jl:83) -- an exact LAPACK SVD instead of a randomized one."""
    U, s, Vt = np.linalg.svd(np.asarray(X, dtype=np.float64), full_matrices=False)
    T = X.dtype.type
    return U[:, :k].astype(T), s[:k].astype(T), Vt[:k].T.astype(T)


def rsvd(X, k, seed=0, ctx: Context | None = None, power_iters=0):
    """rsvd(X, k) -> (U, s, V) on the device (Context.rsvd)."""
    own = ctx is None
    if own:
        ctx = Context(X.dtype.type, X.shape[0], X.shape[1], k)
        ctx.set_X(np.asfortranarray(X))
    try:
        return ctx.rsvd(seed, power_iters=power_iters)
    finally:
        if own:
            ctx.close()


def spa(X, k, ctx: Context | None = None, return_anchors=False, warm_sweeps=16, return_info=False):
    """spa(X, k) -> (W, H) (src/spa.jl:38-63): W = X[:, anchors], H the non-negative least-squares fit (Context.spa_init)."""
    T = X.dtype.type
    p, n = X.shape
    own = ctx is None
    if own:
        ctx = Context(T, p, n, k)
        ctx.set_X(np.asfortranarray(X))
    try:
        anchors, unsolved = ctx.spa_init(warm_sweeps=warm_sweeps)
        W = np.empty((p, k), dtype=T, order="F")
        H = np.empty((k, n), dtype=T, order="F")
        ctx.get_factors(W, H)
    finally:
        if own:
            ctx.close()
    if return_info:
        return W, H, {"anchors": anchors, "unsolved": unsolved}
    return (W, H, anchors) if return_anchors else (W, H)


def nndsvd(X, k, zeroh=False, variant="std", initdata=None, seed=0, ctx: Context | None = None, power_iters=0):
    """nndsvd(X, k; zeroh, variant, initdata) (src/initialization.jl:74-101): the SVD comes from `initdata` = (U, s, V) or,
    like the reference's default, from the randomized rsvd(X, k) -- run on the device, its result never leaves it;
    _nndsvd! runs on the device."""
    T = X.dtype.type
    p, n = X.shape
    if variant not in ("std", "a", "ar"):
        raise ArgumentError("Invalid value for variant")
    own = ctx is None
    if own:
        ctx = Context(T, p, n, k)
        if variant != "std" or initdata is None:
            ctx.set_X(np.asfortranarray(X))
    try:
        if initdata is None:
            ctx.rsvd(seed, download=False, power_iters=power_iters)
            U = s = V = None
        else:
            U, s, V = initdata
        ctx.nndsvd_init(U, s, V, variant=variant, zeroh=zeroh, seed=seed)
        W = np.empty((p, k), dtype=T, order="F")
        H = np.empty((k, n), dtype=T, order="F")
        ctx.get_factors(W, H)
    finally:
        if own:
            ctx.close()
    return W, H


def nnmf(X, k, init="nndsvdar", alg="greedycd", maxiter=100, tol=None, replicates=1, W0=None, H0=None,
         update_H=True, verbose=False, rng=None, track_objective=False, seed=None, initdata=None):
    """nnmf(X, k; init=:nndsvdar, alg=:greedycd, ...) (src/interf.jl:3-83) with the reference's own defaults.

    alg in {greedycd, cd, multmse, multdiv, projals, alspgrad, spa}; init in {nndsvd, nndsvda, nndsvdar, spa, random, custom}
    (alg = spa requires init = spa, src/interf.jl:73-77).

    The NNDSVD initialisers always run on the device front end (randomized SVD + _nndsvd! next to the resident X; the one
    uniform per component of :nndsvdar comes from Philox keyed by `seed`, default 0 -- Julia's stream cannot be reproduced).
    For init in {random, custom}: seed=None draws on the host from `rng` (NumPy) and W0 / H0 are updated IN PLACE like
    the reference does; seed=int uses the device front end -- X is uploaded first and checked for negatives there, the
    random start and the replicate restarts are drawn by nmfx_randinit (Philox4x32-10), and only the winning replicate's
    factors come back (custom W0 / H0 are updated in place there too when they are column-major arrays of X's element type)."""
    T = X.dtype.type
    # host-checkable arguments first (no device needed to reject them)
    if init not in ("nndsvd", "nndsvda", "nndsvdar", "random", "custom", "spa"):
        raise ArgumentError("Invalid value for init.")
    if alg not in _ALGS + ("spa",):
        raise ArgumentError("Invalid algorithm.")
    if alg == "spa" and init != "spa":
        raise ArgumentError("Invalid value for init, use :spa instead.")
    if seed is not None or init in ("nndsvd", "nndsvda", "nndsvdar", "spa"):
        return _nnmf_device(X, k, init, alg, maxiter, tol, replicates, W0, H0, update_H, verbose,
                            int(0 if seed is None else seed), initdata)
    if not (np.issubdtype(X.dtype, np.floating) and np.all(X >= 0)):
        raise ArgumentError("The elements of X must be non-negative.")
    p, n = X.shape
    if not k <= min(p, n):
        raise ArgumentError("The value of k should not exceed min(size(X)).")
    if not replicates >= 1:
        raise ArgumentError("The value of replicates must be positive.")
    if not update_H and init != "custom":
        warnings.warn("Only W will be updated.")
    tol = float(np.cbrt(_eps(T) / 100)) if tol is None else tol      # src/interf.jl:8
    if init == "custom":
        if W0 is None or H0 is None:
            raise ArgumentError("To use :custom initialization, set W0 and H0.")
        if not np.all(W0 >= 0):
            raise ArgumentError("The elements of W0 must be non-negative.")
        if W0.shape != (p, k):
            raise ArgumentError("Invalid size for W0.")
        if not np.all(H0 >= 0):
            raise ArgumentError("The elements of H0 must be non-negative.")
        if H0.shape != (k, n):
            raise ArgumentError("Invalid size for H0.")
    elif W0 is not None or H0 is not None:
        warnings.warn("Ignore W0 and H0 except for :custom initialization.")
    initH = alg != "projals"                                          # src/interf.jl:39
    if init == "random":
        W, H = randinit(X, k, zeroh=not initH, normalize=True, rng=rng)
    elif init == "custom":
        W, H = W0, H0
    else:
        raise ArgumentError("Invalid value for init.")
    W = np.asfortranarray(W, dtype=T)
    H = np.asfortranarray(H, dtype=T)
    inst = _alg_instance(T, alg, maxiter, tol, verbose, update_H)                 # src/interf.jl:61-79
    # solve_replicates! (src/interf.jl:85-101): X is uploaded once and shared by every replicate
    with Context(T, p, n, k) as ctx:
        ctx.set_X(X)
        ret = solve(inst, X, W, H, ctx=ctx, track_objective=track_objective or verbose)
        minobjv = ret.objvalue
        for _ in range(2, replicates + 1):
            Wr, Hr = randinit(X, k, zeroh=not initH, normalize=True, rng=rng)
            tmp = solve(inst, X, Wr, Hr, ctx=ctx)
            if minobjv > tmp.objvalue:
                ret, minobjv = tmp, tmp.objvalue
    return ret


def _alg_instance(T, alg, maxiter, tol, verbose, update_H):
    if alg == "projals":
        return ProjectedALS(T, maxiter=maxiter, tol=tol, verbose=verbose, update_H=update_H)
    if alg == "alspgrad":
        return ALSPGrad(T, maxiter=maxiter, tol=tol, verbose=verbose, update_H=update_H)
    if alg == "multmse":
        return MultUpdate(T, obj="mse", maxiter=maxiter, tol=tol, verbose=verbose, update_H=update_H)
    if alg == "multdiv":
        return MultUpdate(T, obj="div", maxiter=maxiter, tol=tol, verbose=verbose, update_H=update_H)
    if alg == "cd":
        return CoordinateDescent(T, maxiter=maxiter, tol=tol, verbose=verbose, update_H=update_H)
    if alg == "greedycd":
        return GreedyCD(T, maxiter=maxiter, tol=tol, verbose=verbose, update_H=update_H)
    if alg == "spa":
        return SPA(T, obj="mse")
    raise ArgumentError("Invalid algorithm.")


def _nnmf_device(X, k, init, alg, maxiter, tol, replicates, W0, H0, update_H, verbose, seed, initdata=None):
    """nnmf with the front end on the device: same checks, same order, same messages as src/interf.jl:15-101."""
    T = X.dtype.type
    if not np.issubdtype(X.dtype, np.floating):
        raise ArgumentError("The elements of X must be non-negative.")
    p, n = X.shape
    if not k <= min(p, n):
        raise ArgumentError("The value of k should not exceed min(size(X)).")
    if not replicates >= 1:
        raise ArgumentError("The value of replicates must be positive.")
    if not update_H and init != "custom":
        warnings.warn("Only W will be updated.")
    tol = float(np.cbrt(_eps(T) / 100)) if tol is None else tol
    if init == "custom":
        if W0 is None or H0 is None:
            raise ArgumentError("To use :custom initialization, set W0 and H0.")
        if W0.shape != (p, k):
            raise ArgumentError("Invalid size for W0.")
        if H0.shape != (k, n):
            raise ArgumentError("Invalid size for H0.")
    elif init not in ("random", "nndsvd", "nndsvda", "nndsvdar", "spa"):
        raise ArgumentError("Invalid value for init.")
    elif W0 is not None or H0 is not None:
        warnings.warn("Ignore W0 and H0 except for :custom initialization.")
    initH = alg != "projals"
    inst = _alg_instance(T, alg, maxiter, tol, verbose, update_H)
    with Context(T, p, n, k) as ctx:
        ctx.set_X(np.asfortranarray(X))
        if not ctx.check_nonneg(0):
            raise ArgumentError("The elements of X must be non-negative.")
        if init == "custom":
            # like the reference (and the host path above) the caller's W0 / H0 are updated IN PLACE when they already are
            # column-major arrays of the element type; anything else is converted (then the result lives in the converted copy)
            W = W0 if (isinstance(W0, np.ndarray) and W0.dtype == T and W0.flags.f_contiguous) else np.asfortranarray(W0, dtype=T).copy(order="F")
            H = H0 if (isinstance(H0, np.ndarray) and H0.dtype == T and H0.flags.f_contiguous) else np.asfortranarray(H0, dtype=T).copy(order="F")
            ctx.set_factors(W, H)
            if not ctx.check_nonneg(1):
                raise ArgumentError("The elements of W0 must be non-negative.")
            if not ctx.check_nonneg(2):
                raise ArgumentError("The elements of H0 must be non-negative.")
        else:
            W = np.empty((p, k), dtype=T, order="F")
            H = np.empty((k, n), dtype=T, order="F")
            if init == "random":
                ctx.randinit(seed, normalize=True, zeroh=not initH)
            elif init == "spa":                                          # src/interf.jl:50-51
                _, unsolved = ctx.spa_init()
                if unsolved > 0:                                         # like NMFX.jl's spa!: the reference's fnnls always returns the minimiser
                    import warnings
                    warnings.warn(f"spa: the active-set least-squares solve of {unsolved} column(s) of H stopped at its iteration cap or "
                                  "on a Cholesky breakdown; those columns hold a feasible but not optimal H", RuntimeWarning, stacklevel=2)
            else:                                                        # src/interf.jl:44-49
                if initdata is None:
                    ctx.rsvd(seed, download=False)                       # rsvd(X, k), src/initialization.jl:83
                    U = s = V = None
                else:
                    U, s, V = initdata
                ctx.nndsvd_init(U, s, V, variant={"nndsvd": "std", "nndsvda": "a", "nndsvdar": "ar"}[init], zeroh=not initH, seed=seed)
            ctx.get_factors(W, H)
        if verbose or isinstance(inst, SPA):
            # the verbose table needs the per-iteration trace of every replicate: drive solve_replicates! from the host
            # (same draws, same winner rule), X stays resident
            ret, best = solve(inst, X, W, H, ctx=ctx), 1
            for r in range(2, replicates + 1):
                ctx.randinit(seed + r - 1, normalize=True, zeroh=not initH)
                Wr = np.empty((p, k), dtype=T, order="F")
                Hr = np.empty((k, n), dtype=T, order="F")
                ctx.get_factors(Wr, Hr)
                tmp = solve(inst, X, Wr, Hr, ctx=ctx)
                if ret.objvalue > tmp.objvalue:
                    ret, best = tmp, r
            ret.info["best_replicate"] = best
            return ret
        opts = make_opts(T, **inst._opts())
        res, best = ctx.solve_replicates(inst._alg(), opts, replicates, seed, not initH, W, H)
    out = _result(T, W, H, res, None)
    out.info["best_replicate"] = best
    return out
