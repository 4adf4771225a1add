// chol.hpp -- k x k symmetric positive-definite kernels for ProjectedALS
// (potrf!/potrs!/potri! call sites: src/utils.jl:63-84, used by src/projals.jl:94,102).
//
//   potrf_upper_kernel : A = U'U in place (upper triangle), one workgroup, right-looking.
//                        Same per-element operation order as LAPACK's unblocked potf2
//                        (contributions of rows l = 0,1,... subtracted in ascending l).
//   trtri_upper_kernel : Uinv = inv(U), one wave per column (back-substitution on e_j).
// inv(A) = Uinv * Uinv' (what potri! forms) and the solves run through the MFMA GEMM.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace nmfx {

__device__ __forceinline__ float nmfx_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double nmfx_sqrt(double x) { return sqrt(x); }

// A: k x k leading block of a K-ld column-major matrix.  On a non-positive pivot sets
// ctrl->status = NOT_POSDEF (3) and ctrl->done = 1 (PosDefException of potrf!).
template <typename T>
__global__ __launch_bounds__(1024) void potrf_upper_kernel(T *A, int64_t ld, int k, Ctrl *ctrl, int posdef_status) {
    if (ctrl != nullptr && ctrl->done) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char chol_smem[];
    T *rowj = reinterpret_cast<T *>(chol_smem);   // k entries: scaled row j of U
    // the flag lives behind rowj in the SAME dynamic region (a static __shared__ in front of it would
    // shift the dynamic base off 16-byte alignment; guide G17)
    int &fail = *reinterpret_cast<int *>(chol_smem + ((size_t)k * sizeof(T) + 15) / 16 * 16);
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) fail = 0;
    __syncthreads();
    for (int j = 0; j < k; ++j) {
        if (tid == 0) {
            const T d = A[j + (int64_t)j * ld];
            if (!(d > (T)0)) fail = 1;
            else A[j + (int64_t)j * ld] = nmfx_sqrt(d);
        }
        __syncthreads();
        if (fail) break;
        const T dj = A[j + (int64_t)j * ld];
        for (int c = j + 1 + tid; c < k; c += nt) {
            const T v = A[j + (int64_t)c * ld] / dj;
            A[j + (int64_t)c * ld] = v;
            rowj[c] = v;
        }
        __syncthreads();
        const int m = k - j - 1;
        for (int idx = tid; idx < m * m; idx += nt) {
            const int r = j + 1 + idx % m, c = j + 1 + idx / m;
            if (r <= c) A[r + (int64_t)c * ld] -= rowj[r] * rowj[c];
        }
        __syncthreads();
    }
    if (fail && tid == 0 && ctrl != nullptr) {
        ctrl->status = posdef_status;
        ctrl->done = 1;
    }
}

// Uinv (zero-initialised K x K, ld) <- inverse of the upper-triangular k x k factor U.  grid = k waves.
template <typename T>
__global__ __launch_bounds__(64) void trtri_upper_kernel(const T *U, T *Uinv, int64_t ld, int k, const int *done) {
    NMFX_DONE_GUARD(done);
    extern __shared__ __attribute__((aligned(16))) unsigned char chol_smem[];
    T *x = reinterpret_cast<T *>(chol_smem);   // k entries
    const int j = blockIdx.x, lane = threadIdx.x;
    if (lane == 0) x[j] = (T)1 / U[j + (int64_t)j * ld];
    __syncthreads();
    for (int i = j - 1; i >= 0; --i) {
        T s = (T)0;
        for (int l = i + 1 + lane; l <= j; l += 64) s += U[i + (int64_t)l * ld] * x[l];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) x[i] = -s / U[i + (int64_t)i * ld];
        __syncthreads();
    }
    for (int i = lane; i <= j; i += 64) Uinv[i + (int64_t)j * ld] = x[i];
}

}  // namespace nmfx
