// chol.hpp -- k x k symmetric positive-definite kernels for ProjectedALS
// (potrf!/potrs!/potri! call sites: src/utils.jl:63-84, used by src/projals.jl:94,102).
//
//   potrf_upper_kernel : A = U'U in place (upper triangle).  One workgroup, blocked right-looking
//                        (the LAPACK potrf structure): the NB x NB diagonal block is factored in
//                        registers by one wave (lane = column, cross-lane shuffles), the row panel is
//                        solved one column per thread against the LDS copy of the block, the trailing
//                        update A22 -= R'R runs out of the LDS copy of the panel.  No serial chain ever
//                        goes through global memory (the first version did: 1.6 ms at k=256).
//   trtri_upper_kernel : Uinv = inv(U), one wave per column (back-substitution on e_j), with the rows of U
//                        staged through LDS 32 at a time so the serial recurrence only touches LDS.
// inv(A) = Uinv * Uinv' (what potri! forms) and the solves run through the MFMA GEMM.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace nmfx {

__device__ __forceinline__ float nmfx_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double nmfx_sqrt(double x) { return sqrt(x); }

template <typename T> __device__ __forceinline__ T wave_bcast(T v, int src) { return __shfl(v, src, 64); }

// A: k x k leading block of a column-major matrix with leading dimension ld.  On a non-positive pivot sets
// ctrl->status = posdef_status and ctrl->done = 1 (PosDefException of potrf!, src/utils.jl:68,78).
// Dynamic LDS: NB*NB (diagonal block) + NB*k (row panel) elements of T + 16 bytes.
template <typename T, int NB>
__global__ __launch_bounds__(1024) void potrf_upper_kernel(T *A, int64_t ld, int k, Ctrl *ctrl, int posdef_status) {
    if (ctrl != nullptr && ctrl->done) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char chol_smem[];
    T *U11 = reinterpret_cast<T *>(chol_smem);          // U11[l*NB + i] = U(jb+l, jb+i)
    T *Rp = U11 + NB * NB;                              // Rp[l*k + c]   = U(jb+l, jb+nb+c)
    int *failp = reinterpret_cast<int *>(chol_smem + (((size_t)(NB * NB + NB * (size_t)k) * sizeof(T) + 15) / 16) * 16);
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) *failp = 0;
    __syncthreads();
    for (int jb = 0; jb < k; jb += NB) {
        const int nb = (k - jb < NB) ? (k - jb) : NB;
        const int m = k - jb - nb;
        if (wave == 0) {
            // lane c holds column c of the diagonal block (rows 0..c)
            T col[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r)
                col[r] = (lane < nb && r <= lane) ? A[(jb + r) + (int64_t)(jb + lane) * ld] : (T)0;
            bool bad = false;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (j < nb && !bad) {
                    const T d = wave_bcast(col[j], j);
                    if (!(d > (T)0)) {
                        bad = true;
                    } else {
                        const T dj = nmfx_sqrt(d);
                        if (lane == j) col[j] = dj;
                        else if (lane > j) col[j] = col[j] / dj;
#pragma unroll
                        for (int r = j + 1; r < NB; ++r) {
                            const T ujr = wave_bcast(col[j], r);
                            if (lane >= r) col[r] -= ujr * col[j];
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < NB; ++r)
                if (lane < nb && r <= lane) {
                    A[(jb + r) + (int64_t)(jb + lane) * ld] = col[r];
                    U11[r * NB + lane] = col[r];
                }
            if (bad && lane == 0) *failp = 1;
        }
        __syncthreads();
        if (*failp) break;
        // row panel: solve U11' R' = R, one column per thread
        for (int c = tid; c < m; c += nt) {
            T x[NB];
            T *colp = A + jb + (int64_t)(jb + nb + c) * ld;
#pragma unroll
            for (int l = 0; l < NB; ++l) x[l] = (l < nb) ? colp[l] : (T)0;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if (i < nb) {
                    T s = x[i];
#pragma unroll
                    for (int l = 0; l < i; ++l) s -= U11[l * NB + i] * x[l];
                    x[i] = s / U11[i * NB + i];
                }
            }
#pragma unroll
            for (int l = 0; l < NB; ++l)
                if (l < nb) {
                    colp[l] = x[l];
                    Rp[l * k + c] = x[l];
                }
        }
        __syncthreads();
        // trailing update A22 -= R'R (upper part)
        for (int idx = tid; idx < m * m; idx += nt) {
            const int r = idx % m, c = idx / m;
            if (r <= c) {
                T s = (T)0;
                for (int l = 0; l < nb; ++l) s += Rp[l * k + r] * Rp[l * k + c];
                A[(jb + nb + r) + (int64_t)(jb + nb + c) * ld] -= s;
            }
        }
        __syncthreads();
    }
    if (*failp && tid == 0 && ctrl != nullptr) {
        ctrl->status = posdef_status;
        ctrl->done = 1;
    }
}

// Uinv (zero-initialised, ld) <- inverse of the upper-triangular k x k factor U.  grid = k workgroups of one
// wave; workgroup j solves U x = e_j from the bottom up.  Rows are processed in blocks of RB: the part of the
// dot products that only needs already-known x (columns beyond the block) is done lane-parallel straight from
// global memory, the block's own RB x RB triangle is staged in LDS for the short serial recurrence.
// Dynamic LDS: k + RB*RB elements of T (+ RB for the partial sums).
template <typename T, int RB>
__global__ __launch_bounds__(64) void trtri_upper_kernel(const T *U, T *Uinv, int64_t ld, int k, const int *done) {
    NMFX_DONE_GUARD(done);
    extern __shared__ __attribute__((aligned(16))) unsigned char chol_smem[];
    T *x = reinterpret_cast<T *>(chol_smem);   // k entries
    T *tri = x + ((k + 3) / 4) * 4;            // RB x RB: tri[i*RB + l] = U(ib+i, ib+l)
    T *part = tri + RB * RB;                   // RB partial sums
    const int j = blockIdx.x, lane = threadIdx.x;
    for (int i = lane; i < k; i += 64) x[i] = (T)0;
    __syncthreads();
    // last (partial) block ends at row j
    for (int ie = j; ie >= 0; ie -= RB) {
        const int ib = (ie - RB + 1 > 0) ? ie - RB + 1 : 0;   // rows ib..ie
        const int nr = ie - ib + 1;
        // stage the triangle U(ib..ie, ib..ie)
        for (int e = lane; e < nr * nr; e += 64) {
            const int i = e % nr, l = e / nr;
            tri[i * RB + l] = (l >= i) ? U[(ib + i) + (int64_t)(ib + l) * ld] : (T)0;
        }
        // partial sums over the known part: part[i] = sum_{l = ie+1..j} U(ib+i, l) x[l]
        {
            const int i = lane % RB, h = lane / RB;   // RB = 32: two half-waves split the l range
            T s = (T)0;
            if (i < nr)
                for (int l = ie + 1 + h; l <= j; l += 64 / RB) s += U[(ib + i) + (int64_t)l * ld] * x[l];
            s += __shfl_down(s, RB, 64);
            if (lane < RB) part[lane] = s;
        }
        __syncthreads();
        // serial recurrence inside the block (bottom row first), lane-parallel dot over the block's columns
        for (int i = nr - 1; i >= 0; --i) {
            T s = (T)0;
            const int l = lane;
            if (l > i && l < nr) s = tri[i * RB + l] * x[ib + l];
            for (int off = 16; off > 0; off >>= 1) s += __shfl_down(s, off, 64);   // RB = 32 columns live in lanes 0..31
            if (lane == 0) {
                const T rhs = ((ib + i) == j) ? (T)1 : (T)0;
                x[ib + i] = (rhs - part[i] - s) / tri[i * RB + i];
            }
            __syncthreads();
        }
    }
    for (int i = lane; i <= j; i += 64) Uinv[i + (int64_t)j * ld] = x[i];
}

}  // namespace nmfx
