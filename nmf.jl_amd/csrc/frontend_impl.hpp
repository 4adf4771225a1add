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

// ---------------------------------------------------------------------------
// NNDSVD from a given truncated SVD (src/initialization.jl:26-101): the part of nndsvd() behind `U, s, V = ...`.  The SVD
// itself (RandomizedLinAlg.rsvd, or the caller's `initdata`) stays with the host; _nndsvd! -- two column-norm passes and
// one elementwise fill over the p x k and n x k factors -- runs here, next to the resident X whose mean the :a / :ar
// variants need.
// ---------------------------------------------------------------------------
// out[2*j + {0,1}] = sum of squares of the positive / non-positive entries of column j (posnegnorm, :103-115); one block per column
// (component j, sample i) lives at A[j*cs + i*ss]: host-uploaded U / V are column-major (cs = ld, ss = 1); the resident V' of
// rsvd is k x n like H (cs = 1, ss = K)
template <typename T> __global__ void posneg_sumsq_kernel(const T *A, int64_t rows, int64_t cs, int64_t ss, double *out) {
    __shared__ double sp[4], sn[4];
    const T *col = A + (int64_t)blockIdx.x * cs;
    double pn = 0.0, nn = 0.0;
    for (int64_t i = threadIdx.x; i < rows; i += blockDim.x) {
        const T x = col[i * ss];
        const double q = (double)(T)(x * x);
        if (x > (T)0) pn += q; else nn += q;
    }
    for (int off = 32; off > 0; off >>= 1) { pn += __shfl_down(pn, off, 64); nn += __shfl_down(nn, off, 64); }
    if ((threadIdx.x & 63) == 0) { sp[threadIdx.x >> 6] = pn; sn[threadIdx.x >> 6] = nn; }
    __syncthreads();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = sp[0] + sp[1] + sp[2] + sp[3]; out[2 * blockIdx.x + 1] = sn[0] + sn[1] + sn[2] + sn[3]; }
}

// partial[block] = sum of the logical rows x cols block of X (for mean(X), :41-42)
template <typename T> __global__ void sum_block_kernel(const T *A, int64_t rows, int64_t cols, int64_t ld, double *partial) {
    __shared__ double sm[4];
    double s = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < rows * cols; e += (int64_t)gridDim.x * blockDim.x)
        s += (double)A[(e % rows) + (e / rows) * ld];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// per component j (:44-72): which sign pattern wins, the two scale factors and the fill value.
// coef[4*j + {0,1,2,3}] = { cW, cH, sign (+1 / -1), vj }
template <typename T>
__global__ void nndsvd_coef_kernel(const double *unorm, const double *vnorm, const T *sv, const double *xsum, int nsum, double count,
                                   int variant, uint64_t seed, int k, T *coef) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    double tot = 0.0;
    for (int i = 0; i < nsum; ++i) tot += xsum[i];
    const double mean = tot / count;
    const T v0 = (variant == 0) ? (T)0 : (variant == 1) ? (T)mean : (T)(mean * 0.01);   // :41-42
    const T xp = sqrt((T)unorm[2 * j]), xn = sqrt((T)unorm[2 * j + 1]);                   // posnegnorm in T
    const T yp = sqrt((T)vnorm[2 * j]), yn = sqrt((T)vnorm[2 * j + 1]);
    const T mp = xp * yp, mn = xn * yn;                                                    // :49-50
    T vj = v0;
    if (variant == 2) {                                                                    // :52-55  vj *= rand(T)
        uint32_t w[4];
        philox4x32_10((uint32_t)j, 0u, 2u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
        vj = vj * philox_u01<T>(w);
    }
    if (mp >= mn) {                                                                        // :58-61
        const T ss = sqrt(sv[j] * mp);
        coef[4 * j] = ss / xp; coef[4 * j + 1] = ss / yp; coef[4 * j + 2] = (T)1;
    } else {                                                                               // :62-65
        const T ss = sqrt(sv[j] * mn);
        coef[4 * j] = ss / xn; coef[4 * j + 1] = ss / yn; coef[4 * j + 2] = (T)-1;
    }
    coef[4 * j + 3] = vj;
}

// scalepos! / scaleneg! (:117-137): dst(i, j) = x > 0 ? x*c : v0   or   x < 0 ? -(x*c) : v0.
// src element (sample i, component j) at src[j*cs + i*ss]; dst element (i, j) at dst[i*ds_i + j*ds_j] (W: 1, P;  H from V: K, 1)
template <typename T>
__global__ void nndsvd_fill_kernel(const T *src, int64_t rows, int64_t cs, int64_t ss, int k, const T *coef, int which, T *dst,
                                   int64_t ds_i, int64_t ds_j) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * k) return;
    const int64_t i = e % rows;
    const int j = (int)(e / rows);
    const T x = src[(int64_t)j * cs + i * ss];
    const T c = coef[4 * j + which], sg = coef[4 * j + 2], v0 = coef[4 * j + 3];
    T y;
    if (sg > (T)0) y = (x > (T)0) ? x * c : v0;
    else y = (x < (T)0) ? -(x * c) : v0;
    dst[i * ds_i + (int64_t)j * ds_j] = y;
}

template <typename T>
void Solver<T>::nndsvd_core(const T *Ud, int64_t ucs, int64_t uss, const T *Vd, int64_t vcs, int64_t vss, const T *sd, T *coef,
                            int variant, bool zeroh, uint64_t seed, int64_t n_total) {
    if (variant < 0 || variant > 2) throw StatusError{NMFX_ERR_BAD_ARG, "Invalid value for variant"};
    if (variant != 0 && !have_X) throw StatusError{NMFX_ERR_STATE, "variants :a / :ar need mean(X): upload X first (nmfx_set_X)"};
    if (n_total < n) throw StatusError{NMFX_ERR_BAD_ARG, "n_total must be the global column count (>= n_local)"};
    const int nsum = 1024;
    nd_scratch.ensure((size_t)4 * K + nsum + 4096);
    double *unorm = nd_scratch.p, *vnorm = nd_scratch.p + 2 * K, *xsum = nd_scratch.p + 4 * K;
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipMemsetAsync(W[i].p, 0, W[i].count * sizeof(T), stream));
        HIP_TRY(hipMemsetAsync(H[i].p, 0, H[i].count * sizeof(T), stream));
    }
    hipLaunchKernelGGL(posneg_sumsq_kernel<T>, dim3((unsigned)k), dim3(256), 0, stream, Ud, p, ucs, uss, unorm);
    hipLaunchKernelGGL(posneg_sumsq_kernel<T>, dim3((unsigned)k), dim3(256), 0, stream, Vd, n, vcs, vss, vnorm);
    HIP_TRY(hipMemsetAsync(xsum, 0, nsum * sizeof(double), stream));
    if (variant != 0) hipLaunchKernelGGL(sum_block_kernel<T>, dim3(nsum), dim3(256), 0, stream, X.p, p, n, P, xsum);
    if (sharded()) {   // V rows (= columns of X, H) are sharded: the norms of V's columns and sum(X) are global quantities
        comm->group_start();
        comm->all_reduce(vnorm, (size_t)2 * k, CT_F64, false, stream);
        comm->all_reduce(xsum, (size_t)nsum, CT_F64, false, stream);
        comm->group_end();
    }
    hipLaunchKernelGGL(nndsvd_coef_kernel<T>, dim3((unsigned)((k + 63) / 64)), dim3(64), 0, stream, unorm, vnorm, sd, xsum, nsum,
                       (double)p * (double)n_total, variant, seed, (int)k, coef);
    hipLaunchKernelGGL(nndsvd_fill_kernel<T>, dim3((unsigned)((p * k + 255) / 256)), dim3(256), 0, stream, Ud, p, ucs, uss, (int)k, coef, 0,
                       W[0].p, (int64_t)1, P);
    if (!zeroh)
        hipLaunchKernelGGL(nndsvd_fill_kernel<T>, dim3((unsigned)((n * k + 255) / 256)), dim3(256), 0, stream, Vd, n, vcs, vss, (int)k, coef,
                           1, H[0].p, K, (int64_t)1);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));
    wcur = hcur = 0;
    have_F = true;
}

template <typename T>
void Solver<T>::nndsvd_init(const void *U_host, const void *s_host, const void *V_host, int variant, bool zeroh, uint64_t seed,
                            int64_t n_total) {
    HIP_TRY(hipSetDevice(device));
    if (U_host == nullptr) {   // the resident U, s, V' left by nmfx_rsvd_finish
        if (rsvd_ready < 2) throw StatusError{NMFX_ERR_STATE, "no resident SVD: call nmfx_rsvd_begin / nmfx_rsvd_finish first"};
        nndsvd_core(work[5].p, P, 1, work[7].p, 1, K, work[6].p, work[6].p + K, variant, zeroh, seed, n_total);
        return;
    }
    rsvd_ready = 0;            // the scratch buffers are shared with rsvd
    work[4].ensure((size_t)p * k);
    work[5].ensure((size_t)std::max<int64_t>(n * k, 1));
    work[6].ensure((size_t)5 * K);
    T *Ud = work[4].p, *Vd = work[5].p, *sd = work[6].p, *coef = work[6].p + K;
    HIP_TRY(hipMemcpyAsync(Ud, U_host, (size_t)p * k * sizeof(T), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(Vd, V_host, (size_t)n * k * sizeof(T), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(sd, s_host, (size_t)k * sizeof(T), hipMemcpyHostToDevice, stream));
    nndsvd_core(Ud, p, 1, Vd, n, 1, sd, coef, variant, zeroh, seed, n_total);
}

// solve_replicates! (src/interf.jl:85-101): replicate 1 starts from the caller's W, H; replicates 2..R from fresh
// randinit(normalize = true, zeroh) draws (seed + r - 1); the result with the smallest objective is kept
// (`if minobjv > tmp.objvalue`, strictly smaller, so ties keep the earlier one).  X stays resident; the best factors are
// parked in two device buffers and only the winner crosses PCIe.
template <typename T>
void Solver<T>::solve_replicates(int alg, const nmfx_opts &o, int replicates, uint64_t seed, bool zeroh, int64_t h_col_offset,
                                 void *W_host, void *H_host, nmfx_result *out, int *best) {
    if (replicates < 1) throw StatusError{NMFX_ERR_BAD_ARG, "The value of replicates must be positive."};
    if (replicas_mode()) {
        // ---- the replicates dealt out over the ranks (NMFX_COMM_REPLICAS, include/nmfx.h) ------------------------------------
        const int G = nranks, per = (replicates + G - 1) / G;
        std::vector<nmfx_result> mine((size_t)per), all((size_t)per * G);
        for (auto &m : mine) { std::memset(&m, 0, sizeof m); m.objvalue = std::numeric_limits<double>::infinity(); m.niters = -1; }   // niters = -1: no such replicate
        Wbest.ensure(W[0].count);
        Hbest.ensure(H[0].count);
        auto park = [&] {
            HIP_TRY(hipMemcpyAsync(Wbest.p, W[wcur].p, W[0].count * sizeof(T), hipMemcpyDeviceToDevice, stream));
            HIP_TRY(hipMemcpyAsync(Hbest.p, H[hcur].p, H[0].count * sizeof(T), hipMemcpyDeviceToDevice, stream));
        };
        // local candidate = the replicate of this rank the global scan could pick: the first one that is strictly smaller than every
        // earlier one of this rank (a NaN objective never wins, except replicate 1, which then wins everything -- on rank 0 it is first)
        double local_best = std::numeric_limits<double>::quiet_NaN();
        bool have_local = false;
        std::string fail_msg;
        for (int i = 0; i < per; ++i) {
            const int r = rank + 1 + i * G;
            if (r > replicates) break;
            if (r == 1) set_factors(W_host, H_host);
            else randinit(seed + (uint64_t)(r - 1), /*normalize=*/true, zeroh, h_col_offset);
            nmfx_result res;
            // A replicate that fails (PosDefException of a ProjectedALS factorisation, a non-finite step size, an argument error) must not
            // take this rank out of the collectives the others are about to enter: its status is recorded (niters = -2), the rank stops
            // computing -- the sequential loop would have thrown here -- and still joins the all-gather; every rank then raises the error
            // of the FIRST failing replicate in replicate order, as the reference's loop does (src/interf.jl:91-98).
            try {
                iterate(alg, o, &res, nullptr);
            } catch (const StatusError &e) {
                std::memset(&res, 0, sizeof res);
                res.niters = -2;
                res.status = e.status;
                res.objvalue = std::numeric_limits<double>::quiet_NaN();
                mine[(size_t)i] = res;
                fail_msg = e.msg;
                break;
            }
            mine[(size_t)i] = res;
            bool take;
            if (!have_local) take = true;
            else if (local_best != local_best) take = rank != 0 && res.objvalue == res.objvalue;   // (rank 0's NaN can only be replicate 1's: it keeps everything)
            else take = local_best > res.objvalue;
            if (take) { local_best = res.objvalue; have_local = true; park(); }
        }
        // every replicate's record to every rank
        const size_t rec = sizeof(nmfx_result), chunk = rec * (size_t)per;
        rep_buf.ensure(chunk * (size_t)(G + 1));
        unsigned char *send = rep_buf.p, *recv = rep_buf.p + chunk;
        HIP_TRY(hipMemcpyAsync(send, mine.data(), chunk, hipMemcpyHostToDevice, stream));
        comm->all_gather(send, recv, chunk, CT_BYTE, stream);
        HIP_TRY(hipMemcpyAsync(all.data(), recv, chunk * (size_t)G, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        // the reference's scan over r = 1 .. R (src/interf.jl:91-98)
        auto rec_of = [&](int r) -> const nmfx_result & { return all[(size_t)((r - 1) % G) * per + (size_t)((r - 1) / G)]; };
        for (int r = 1; r <= replicates; ++r)
            if (rec_of(r).niters == -2) {   // every rank sees the same records: all of them throw, none enters the broadcast
                const int st = rec_of(r).status != 0 ? rec_of(r).status : NMFX_ERR_STATE;
                const bool here = (r - 1) % G == rank && !fail_msg.empty();
                throw StatusError{st, here ? fail_msg : ("replicate " + std::to_string(r) + " failed on rank " + std::to_string((r - 1) % G) +
                                                          (st == NMFX_ERR_NOT_POSDEF ? ": matrix is not positive definite (potrf)" : ""))};
            }
        int best_r = 1;
        nmfx_result best_res = rec_of(1);
        for (int r = 2; r <= replicates; ++r)
            if (best_res.objvalue > rec_of(r).objvalue) { best_res = rec_of(r); best_r = r; }
        const int owner = (best_r - 1) % G;
        if (rank == owner) {
            HIP_TRY(hipMemcpyAsync(W[wcur].p, Wbest.p, W[0].count * sizeof(T), hipMemcpyDeviceToDevice, stream));
            HIP_TRY(hipMemcpyAsync(H[hcur].p, Hbest.p, H[0].count * sizeof(T), hipMemcpyDeviceToDevice, stream));
        }
        comm->broadcast(W[wcur].p, W[0].count * sizeof(T), owner, stream);
        // (H is broadcast ALWAYS: with update_H = false and replicate 1 the winner the other ranks' device H would otherwise keep the
        // randinit draw of their last replicate, and a following nmfx_iterate / nmfx_get_factors would see different contexts; only the
        // copy into the caller's H is skipped, so that it comes back untouched: test/interf.jl:35)
        comm->broadcast(H[hcur].p, H[0].count * sizeof(T), owner, stream);
        have_F = true;
        get_factors(W_host, (o.update_H || best_r != 1) ? H_host : nullptr);
        if (comm) comm->health();
        *out = best_res;
        if (best) *best = best_r;
        return;
    }
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
