"""ctypes binding of libnmfx.so -- the C ABI declared in include/nmfx.h.

There is NO CPU fallback: if the HIP library is missing or no GPU is visible the
import / context creation fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NMFX_LIB", os.path.join(_HERE, "..", "lib", "libnmfx.so"))

F32, F64 = 0, 1
ALG_MULTMSE, ALG_MULTDIV, ALG_PROJALS, ALG_ALSPGRAD, ALG_CD, ALG_GREEDYCD = 0, 1, 2, 3, 4, 5
PREC_FP32, PREC_BF16X3 = 0, 1
OK, ERR_BAD_ARG, ERR_DIM_MISMATCH, ERR_NOT_POSDEF, ERR_ALPHA_NONFINITE, ERR_HIP, ERR_RCCL, ERR_NO_DEVICE, ERR_STATE, ERR_UNSUPPORTED = range(10)
UNIQUE_ID_BYTES = 128
P2P_HANDLE_BYTES = 128

# every symbol include/nmfx.h declares (tests check the library exports all of them)
SYMBOLS = [
    "nmfx_create", "nmfx_destroy", "nmfx_last_error", "nmfx_version", "nmfx_set_X", "nmfx_set_X_device",
    "nmfx_set_factors", "nmfx_get_factors", "nmfx_iterate", "nmfx_solve", "nmfx_alspgrad_subsolve",
    "nmfx_comm_get_unique_id", "nmfx_comm_init", "nmfx_objective", "nmfx_profile_enable", "nmfx_profile_get", "nmfx_set_final_objective",
    "nmfx_device_info", "nmfx_check_nonneg", "nmfx_randinit", "nmfx_solve_replicates", "nmfx_nndsvd", "nmfx_get_iter_trace", "nmfx_rsvd_begin", "nmfx_rsvd_finish",
    "nmfx_local_group_create", "nmfx_local_group_destroy", "nmfx_comm_init_local", "nmfx_comm_set_mode", "nmfx_comm_init_sim", "nmfx_comm_init_p2p", "nmfx_comm_p2p_export", "nmfx_comm_p2p_attach", "nmfx_comm_p2p_stats", "nmfx_pdsolve", "nmfx_pdrsolve", "nmfx_spa_init",
]
COMM_ROW_SHARDED, COMM_REPLICATED_W, COMM_PIPELINED, COMM_REPLICAS = 0, 1, 2, 3


class Opts(C.Structure):
    _fields_ = [("maxiter", C.c_int32), ("update_H", C.c_int32), ("track_objective", C.c_int32),
                ("maxsubiter", C.c_int32), ("traceiter", C.c_int32), ("check_every", C.c_int32),
                ("tol", C.c_double), ("lambda_w", C.c_double), ("lambda_h", C.c_double), ("delta", C.c_double),
                ("tolg", C.c_double), ("beta", C.c_double), ("sigma", C.c_double),
                ("l1_w", C.c_double), ("l2_w", C.c_double), ("l1_h", C.c_double), ("l2_h", C.c_double),
                ("precision", C.c_int32), ("cd_shuffle", C.c_int32), ("pg_refresh", C.c_int32), ("h_solve", C.c_int32), ("stop_sums", C.c_int32), ("reserved0", C.c_int32)]


class CResult(C.Structure):
    _fields_ = [("niters", C.c_int64), ("converged", C.c_int32), ("status", C.c_int32),
                ("objvalue", C.c_double), ("seconds_loop", C.c_double), ("inner_iters", C.c_int64),
                ("backtracks", C.c_int64), ("final_tolg", C.c_double)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("ms_total", C.c_double), ("launches", C.c_int64),
                ("flops", C.c_double), ("bytes", C.c_double)]


_lib = None


def load():
    """Load libnmfx.so (built by __graft_entry__.build()).  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64/librccl
    # (same SONAMEs as /opt/rocm).  Whichever is loaded first serves both; loading /opt/rocm's first and torch's
    # extensions afterwards mixes runtime versions (observed: abort at exit).  If torch is installed, let it load
    # first -- it is only plumbing here (device tensors / torch.distributed in bench.py and the tests).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = os.path.abspath(LIB_PATH)
    if not os.path.exists(path):
        raise ImportError(f"libnmfx.so not found at {path}: build it with `python __graft_entry__.py` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
    lib.nmfx_create.argtypes = [C.POINTER(vp), i32, i64, i64, i64, i32]
    lib.nmfx_destroy.argtypes = [vp]
    lib.nmfx_destroy.restype = None
    lib.nmfx_last_error.argtypes = [vp]
    lib.nmfx_last_error.restype = C.c_char_p
    lib.nmfx_version.restype = C.c_char_p
    lib.nmfx_set_X.argtypes = [vp, vp, i64]
    lib.nmfx_set_X_device.argtypes = [vp, vp, i64]
    lib.nmfx_set_factors.argtypes = [vp, vp, vp]
    lib.nmfx_get_factors.argtypes = [vp, vp, vp]
    lib.nmfx_iterate.argtypes = [vp, i32, C.POINTER(Opts), C.POINTER(CResult), vp]
    lib.nmfx_solve.argtypes = [vp, i32, C.POINTER(Opts), vp, vp, C.POINTER(CResult), vp]
    lib.nmfx_alspgrad_subsolve.argtypes = [vp, i32, C.POINTER(Opts), vp, vp, C.POINTER(CResult)]
    lib.nmfx_comm_get_unique_id.argtypes = [vp]
    lib.nmfx_comm_init.argtypes = [vp, vp, i32, i32]
    lib.nmfx_local_group_create.argtypes = [C.POINTER(vp), i32]
    lib.nmfx_local_group_destroy.argtypes = [vp]
    lib.nmfx_local_group_destroy.restype = None
    lib.nmfx_comm_init_local.argtypes = [vp, vp, i32]
    lib.nmfx_comm_set_mode.argtypes = [vp, i32]
    lib.nmfx_comm_init_sim.argtypes = [vp, i32, i32]
    lib.nmfx_comm_init_p2p.argtypes = [vp, i32, i32]
    lib.nmfx_comm_p2p_export.argtypes = [vp, vp]
    lib.nmfx_comm_p2p_attach.argtypes = [vp, vp]
    lib.nmfx_comm_p2p_stats.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    lib.nmfx_spa_init.argtypes = [vp, i32, vp, C.POINTER(C.c_int64)]
    lib.nmfx_pdsolve.argtypes = [vp, vp, C.c_double, vp, vp, i32]
    lib.nmfx_pdrsolve.argtypes = [vp, vp, vp, C.c_double, vp, i32]
    lib.nmfx_objective.argtypes = [vp, i32, C.POINTER(Opts), C.POINTER(C.c_double)]
    lib.nmfx_check_nonneg.argtypes = [vp, i32, C.POINTER(i32)]
    lib.nmfx_randinit.argtypes = [vp, C.c_uint64, i32, i32, i64]
    lib.nmfx_solve_replicates.argtypes = [vp, i32, C.POINTER(Opts), i32, C.c_uint64, i32, i64, vp, vp, C.POINTER(CResult),
                                          C.POINTER(i32)]
    lib.nmfx_nndsvd.argtypes = [vp, vp, vp, vp, i32, i32, C.c_uint64, i64]
    lib.nmfx_rsvd_begin.argtypes = [vp, C.c_uint64, i64, i32, vp]
    lib.nmfx_rsvd_finish.argtypes = [vp, vp, vp, vp, vp]
    lib.nmfx_get_iter_trace.argtypes = [vp, vp, vp, i32, C.POINTER(i32)]
    lib.nmfx_profile_enable.argtypes = [vp, i32]
    lib.nmfx_set_final_objective.argtypes = [vp, i32]
    lib.nmfx_profile_get.argtypes = [vp, C.POINTER(KernelStat), i32, C.POINTER(i32)]
    lib.nmfx_device_info.argtypes = [i32, C.c_char_p, i32, C.POINTER(i32), C.POINTER(i64)]
    for s in SYMBOLS:
        if getattr(lib, s).restype is C.c_int:
            pass
    _lib = lib
    return lib
