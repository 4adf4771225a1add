"""ctypes binding of oracle/libnmf_oracle.so (the C restatement) -- TEST INFRASTRUCTURE.

Same call surface as oracle/nmf_oracle.py::solve so tests can run both
restatements on the same inputs.  Build with `make -C oracle`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from nmf_oracle import ALG_NAMES, Opts, Result, resolve_opts

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _COpts(C.Structure):
    _fields_ = [("maxiter", C.c_int), ("update_H", C.c_int), ("track_objective", C.c_int),
                ("maxsubiter", C.c_int), ("traceiter", C.c_int), ("_pad", C.c_int),
                ("tol", C.c_double), ("lambda_w", C.c_double), ("lambda_h", C.c_double),
                ("delta", C.c_double), ("tolg", C.c_double), ("beta", C.c_double), ("sigma", C.c_double),
                ("l1_w", C.c_double), ("l2_w", C.c_double), ("l1_h", C.c_double), ("l2_h", C.c_double)]


class _CResult(C.Structure):
    _fields_ = [("niters", C.c_long), ("converged", C.c_int), ("_pad", C.c_int),
                ("objvalue", C.c_double), ("inner_iters", C.c_long), ("backtracks", C.c_long),
                ("final_tolg", C.c_double)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libnmf_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        for sfx in ("f32", "f64"):
            getattr(_LIB, f"nmf_oracle_solve_{sfx}").restype = C.c_int
            getattr(_LIB, f"nmf_oracle_updateh_{sfx}").restype = C.c_long
            getattr(_LIB, f"nmf_oracle_updatew_{sfx}").restype = C.c_long
    return _LIB


def _sfx(dt):
    return "f32" if dt == np.float32 else "f64"


def solve(alg, X, W, H, opts: Opts | None = None) -> Result:
    if isinstance(alg, str):
        alg = ALG_NAMES[alg]
    T = X.dtype.type
    assert X.flags.f_contiguous and W.flags.f_contiguous and H.flags.f_contiguous
    p, n = X.shape
    k = W.shape[1]
    o = resolve_opts(alg, T, opts or Opts())
    co = _COpts(o.maxiter, int(o.update_H), int(o.track_objective), o.maxsubiter, o.traceiter, 0,
                o.tol, o.lambda_w, o.lambda_h, o.delta, o.tolg, float(T(o.beta)), float(T(o.sigma)),
                o.l1_w, o.l2_w, o.l1_h, o.l2_h)
    res = _CResult()
    trace = np.full(o.maxiter + 1, np.nan)
    fn = getattr(lib(), f"nmf_oracle_solve_{_sfx(X.dtype)}")
    st = fn(C.c_int(alg), C.byref(co), C.c_long(p), C.c_long(n), C.c_long(k),
            X.ctypes.data_as(C.c_void_p), W.ctypes.data_as(C.c_void_p), H.ctypes.data_as(C.c_void_p),
            C.byref(res), trace.ctypes.data_as(C.c_void_p))
    if st == 3:
        raise np.linalg.LinAlgError("matrix is not positive definite (PosDefException)")
    if st == 4:
        raise FloatingPointError("alpha is not finite")
    tr = list(trace[: res.niters + 1]) if o.track_objective else []
    return Result(W, H, int(res.niters), bool(res.converged), float(res.objvalue), tr,
                  {"inner": int(res.inner_iters), "backtracks": int(res.backtracks), "final_tolg": res.final_tolg})


def alspgrad_updateh(X, W, H, maxiter=1000, traceiter=20, tolg=None, beta=0.2, sigma=0.01):
    T = H.dtype.type
    tolg = float(T(np.cbrt(np.finfo(T).eps))) if tolg is None else float(T(tolg))
    p, n = X.shape
    k = W.shape[1]
    fn = getattr(lib(), f"nmf_oracle_updateh_{_sfx(X.dtype)}")
    return fn(C.c_long(p), C.c_long(n), C.c_long(k), X.ctypes.data_as(C.c_void_p), W.ctypes.data_as(C.c_void_p),
              H.ctypes.data_as(C.c_void_p), C.c_long(maxiter), C.c_long(traceiter), C.c_double(tolg),
              C.c_double(float(T(beta))), C.c_double(float(T(sigma))))


def alspgrad_updatew(X, W, H, maxiter=1000, traceiter=20, tolg=None, beta=0.2, sigma=0.01):
    T = W.dtype.type
    tolg = float(T(np.cbrt(np.finfo(T).eps))) if tolg is None else float(T(tolg))
    p, n = X.shape
    k = W.shape[1]
    fn = getattr(lib(), f"nmf_oracle_updatew_{_sfx(X.dtype)}")
    return fn(C.c_long(p), C.c_long(n), C.c_long(k), X.ctypes.data_as(C.c_void_p), W.ctypes.data_as(C.c_void_p),
              H.ctypes.data_as(C.c_void_p), C.c_long(maxiter), C.c_long(traceiter), C.c_double(tolg),
              C.c_double(float(T(beta))), C.c_double(float(T(sigma))))


def pdsolve(A, B):
    """pdsolve!(A, x) (src/utils.jl:63-70) through the C restatement; returns inv(A) B."""
    A = np.asfortranarray(A.copy()); B = np.asfortranarray(B.copy())
    fn = getattr(lib(), f"nmf_oracle_pdsolve_{_sfx(A.dtype)}")
    st = fn(C.c_long(A.shape[0]), A.ctypes.data_as(C.c_void_p), C.c_long(B.shape[1]), B.ctypes.data_as(C.c_void_p))
    if st:
        raise np.linalg.LinAlgError("not positive definite")
    return B


def pdrsolve(A, B):
    """pdrsolve!(A, B, x) (src/utils.jl:72-84) through the C restatement; returns A inv(B)."""
    A = np.asfortranarray(A.copy()); B = np.asfortranarray(B.copy())
    Xo = np.zeros_like(A, order="F")
    fn = getattr(lib(), f"nmf_oracle_pdrsolve_{_sfx(A.dtype)}")
    st = fn(C.c_long(A.shape[0]), C.c_long(B.shape[0]), A.ctypes.data_as(C.c_void_p), B.ctypes.data_as(C.c_void_p),
            Xo.ctypes.data_as(C.c_void_p))
    if st:
        raise np.linalg.LinAlgError("not positive definite")
    return Xo


def stop_condition(W, preW, H, preH, tol):
    fn = getattr(lib(), f"nmf_oracle_stop_condition_{_sfx(W.dtype)}")
    p, k = W.shape
    n = H.shape[1]
    return bool(fn(C.c_long(p), C.c_long(n), C.c_long(k), W.ctypes.data_as(C.c_void_p), preW.ctypes.data_as(C.c_void_p),
                   H.ctypes.data_as(C.c_void_p), preH.ctypes.data_as(C.c_void_p), C.c_double(tol)))
