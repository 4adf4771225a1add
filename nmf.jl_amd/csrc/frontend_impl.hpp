// frontend_impl.hpp -- the pieces of nnmf()'s front end that touch the big arrays, done on the device next to the
// resident X:
//   * the non-negativity checks of X / W0 / H0          src/interf.jl:15, 28, 31
//   * randinit(X, k; normalize, zeroh)                  src/initialization.jl:4-17
//   * solve_replicates!: `replicates` restarts on the SAME uploaded X, keep the smallest objective
//                                                       src/interf.jl:85-101
// Julia's `rand` stream (Xoshiro256++) cannot be reproduced outside Julia, so the device generator is
// its own documented counter-based one: Philox4x32-10 keyed by the 64-bit seed, ONE call per matrix element with the
// element's global column-major index as counter -- the numbers do not depend on padding, launch shape or on how the
// columns of H are sharded over GPUs.  tests/philox_ref.py is the NumPy twin the tests compare against bit for bit.
#pragma once
#include "solver.hpp"

namespace nmfx {

__host__ __device__ inline uint32_t philox_mulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }

// Philox4x32-10 (Salmon et al., SC'11): counter c[4], key k[2] -> 4 x 32 random bits
__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = philox_mulhi(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = philox_mulhi(M1, c2), lo1 = M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// U[0,1) like rand(T): f32 from the top 24 bits of word 0; f64 from 53 bits of words 0 (high 27) and 1 (low 26)
template <typename T> __host__ __device__ inline T philox_u01(const uint32_t (&w)[4]);
template <> __host__ __device__ inline float philox_u01<float>(const uint32_t (&w)[4]) { return (float)(w[0] >> 8) * (1.0f / 16777216.0f); }
template <> __host__ __device__ inline double philox_u01<double>(const uint32_t (&w)[4]) {
    return ((double)(w[0] >> 5) * 67108864.0 + (double)(w[1] >> 6)) * (1.0 / 9007199254740992.0);
}

// A(i, j) = U[0,1) for the logical rows x cols block of a column-major array with leading dimension ld (padding untouched).
// Counter = global element index i + (j + col_offset) * rows (64-bit, words 0/1), stream id in word 2.
template <typename T>
__global__ void randfill_kernel(T *A, int64_t rows, int64_t cols, int64_t ld, uint64_t seed, uint32_t stream_id, int64_t col_offset) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * cols) return;
    const int64_t i = e % rows, j = e / rows;
    const uint64_t g = (uint64_t)i + (uint64_t)(j + col_offset) * (uint64_t)rows;
    uint32_t w[4];
    philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), stream_id, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
    A[i + j * ld] = philox_u01<T>(w);
}

// normalize1_cols! (src/utils.jl): every column divided by its sum.  One block per column; the sum is formed in Float64 in a
// fixed order (thread-strided partials, then a fixed tree), the division is done in T.
template <typename T> __global__ void normalize_cols_kernel(T *A, int64_t rows, int64_t ld) {
    __shared__ double sm[4];
    T *col = A + (int64_t)blockIdx.x * ld;
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < rows; i += blockDim.x) s += (double)col[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    const T tot = (T)(sm[0] + sm[1] + sm[2] + sm[3]);
    for (int64_t i = threadIdx.x; i < rows; i += blockDim.x) col[i] = col[i] / tot;
}

// *flag |= 1 if any element of the logical block fails `t >= 0` (negative or NaN, like all(t -> t >= zero(T), X))
template <typename T> __global__ void any_negative_kernel(const T *A, int64_t rows, int64_t cols, int64_t ld, int *flag) {
    const int64_t nvec = rows * cols;
    int bad = 0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nvec; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e % rows, j = e / rows;
        const T t = A[i + j * ld];
        bad |= !(t >= (T)0);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

template <typename T> bool Solver<T>::check_nonneg(int which) {
    HIP_TRY(hipSetDevice(device));
    const T *A;
    int64_t rows, cols, ld;
    if (which == 0) {
        if (!have_X) throw StatusError{NMFX_ERR_STATE, "X has not been uploaded (nmfx_set_X)"};
        A = X.p; rows = p; cols = n; ld = P;
    } else if (which == 1 || which == 2) {
        if (!have_F) throw StatusError{NMFX_ERR_STATE, "W/H have not been uploaded (nmfx_set_factors)"};
        if (which == 1) { A = W[wcur].p; rows = p; cols = k; ld = P; }
        else { A = H[hcur].p; rows = k; cols = n; ld = K; }
    } else {
        throw StatusError{NMFX_ERR_BAD_ARG, "which must be 0 (X), 1 (W) or 2 (H)"};
    }
    flagbuf.ensure(1);
    int *flag = flagbuf.p;
    HIP_TRY(hipMemsetAsync(flag, 0, sizeof(int), stream));
    const int64_t total = rows * cols;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 16 * 1024);
    hipLaunchKernelGGL(any_negative_kernel<T>, dim3(blocks), dim3(256), 0, stream, A, rows, cols, ld, flag);
    HIP_TRY(hipGetLastError());
    int h = 0;
    HIP_TRY(hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return h == 0;
}

template <typename T> void Solver<T>::randinit(uint64_t seed, bool normalize, bool zeroh, int64_t h_col_offset) {
    HIP_TRY(hipSetDevice(device));
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipMemsetAsync(W[i].p, 0, W[i].count * sizeof(T), stream));
        HIP_TRY(hipMemsetAsync(H[i].p, 0, H[i].count * sizeof(T), stream));
    }
    hipLaunchKernelGGL(randfill_kernel<T>, dim3((unsigned)((p * k + 255) / 256)), dim3(256), 0, stream, W[0].p, p, k, P, seed, 0u,
                       (int64_t)0);
    if (normalize) hipLaunchKernelGGL(normalize_cols_kernel<T>, dim3((unsigned)k), dim3(256), 0, stream, W[0].p, p, P);
    if (!zeroh)
        hipLaunchKernelGGL(randfill_kernel<T>, dim3((unsigned)((k * n + 255) / 256)), dim3(256), 0, stream, H[0].p, k, n, K, seed, 1u,
                           h_col_offset);
    HIP_TRY(hipGetLastError());
    wcur = hcur = 0;
    have_F = true;
}

// solve_replicates! (src/interf.jl:85-101): replicate 1 starts from the caller's W, H; replicates 2..R from fresh
// randinit(normalize = true, zeroh) draws (seed + r - 1); the result with the smallest objective is kept
// (`if minobjv > tmp.objvalue`, strictly smaller, so ties keep the earlier one).  X stays resident; the best factors are
// parked in two device buffers and only the winner crosses PCIe.
template <typename T>
void Solver<T>::solve_replicates(int alg, const nmfx_opts &o, int replicates, uint64_t seed, bool zeroh, int64_t h_col_offset,
                                 void *W_host, void *H_host, nmfx_result *out, int *best) {
    if (replicates < 1) throw StatusError{NMFX_ERR_BAD_ARG, "The value of replicates must be positive."};
    set_factors(W_host, H_host);
    nmfx_result res;
    iterate(alg, o, &res, nullptr);
    int best_r = 1;
    nmfx_result best_res = res;
    if (replicates > 1) {
        Wbest.ensure(W[0].count);
        Hbest.ensure(H[0].count);
        auto park = [&] {
            HIP_TRY(hipMemcpyAsync(Wbest.p, W[wcur].p, W[0].count * sizeof(T), hipMemcpyDeviceToDevice, stream));
            HIP_TRY(hipMemcpyAsync(Hbest.p, H[hcur].p, H[0].count * sizeof(T), hipMemcpyDeviceToDevice, stream));
        };
        park();
        for (int r = 2; r <= replicates; ++r) {
            randinit(seed + (uint64_t)(r - 1), /*normalize=*/true, zeroh, h_col_offset);
            iterate(alg, o, &res, nullptr);
            // (a PosDefException / non-finite alpha in any replicate throws out of iterate(), like the reference)
            if (best_res.objvalue > res.objvalue) {
                best_res = res;
                best_r = r;
                park();
            }
        }
        // hand the winner back through the regular download path
        HIP_TRY(hipMemcpyAsync(W[wcur].p, Wbest.p, W[0].count * sizeof(T), hipMemcpyDeviceToDevice, stream));
        HIP_TRY(hipMemcpyAsync(H[hcur].p, Hbest.p, H[0].count * sizeof(T), hipMemcpyDeviceToDevice, stream));
    }
    // update_H == 0 and the winner is replicate 1: H must come back bit-identical -> not even copied
    get_factors(W_host, (o.update_H || best_r != 1) ? H_host : nullptr);
    *out = best_res;
    if (best) *best = best_r;
}

}  // namespace nmfx
