// kernels.hpp -- the HBM-bound passes of the NMF hot path (everything that is not a GEMM):
// split-K slab reduction, stop_condition statistics, convergence check, multdiv
// scalings, objective finalisation.  All reductions use fixed summation orders
// (no float atomics) so a solve is bit-reproducible run to run.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace nmfx {

// Device-resident loop control (replaces the host-side `converged`/`t` of nmf_skeleton!,
// src/common.jl:62-64).  Kernels launched for iterations after `done` are no-ops.
struct Ctrl {
    int done;
    int converged;
    int status;          // nmfx_status raised on device (NOT_POSDEF, ALPHA_NONFINITE)
    int pad;
    long long niters;
    long long inner_iters;
    long long backtracks;
    double tolg;         // ALSPGradUpd.tolg (mutable, src/alspgrad.jl:375-379)
};

#define NMFX_DONE_GUARD(done) \
    if ((done) != nullptr && *reinterpret_cast<const volatile int *>(done) != 0) return

// dst[i] = sum_s src[s*stride + i]  (s ascending: deterministic split-K combine)
template <typename T>
__global__ void reduce_slabs_kernel(T *dst, const T *src, int64_t count, int nslab, int64_t stride,
                                    const int *done) {
    NMFX_DONE_GUARD(done);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    T s = src[i];
    for (int k = 1; k < nslab; ++k) s += src[(int64_t)k * stride + i];
    dst[i] = s;
}

// the same sum, 4 (Float32) / 2 (Float64) consecutive elements per thread as one 16-byte access per slab: a quarter of the load
// instructions and 1 KB per wave-load (count, stride and the base pointers multiples of the vector: the padded operand sizes are)
// (eight slabs' loads in flight, then the adds in slab order: the plain `s += load` loop over a run-time count chained nslab
// dependent memory round trips -- 6.4 us for the 8 slabs of a 256 x 256 Gram)
template <typename T>
__device__ __forceinline__ void reduce_slabs_vec_body(T *dst, const T *src, int64_t nvec, int nslab, int64_t stride, int64_t i, T *dst2 = nullptr) {
    constexpr int V = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(V)));
    if (i >= nvec) return;
    vec_t s = *reinterpret_cast<const vec_t *>(src + i * V);
    for (int k0 = 1; k0 < nslab; k0 += 8) {
        vec_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = (k0 + u < nslab) ? k0 + u : nslab - 1;
            v[u] = *reinterpret_cast<const vec_t *>(src + (int64_t)k * stride + i * V);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + u < nslab) s += v[u];
    }
    *reinterpret_cast<vec_t *>(dst + i * V) = s;
    // (dst2: the same sum into the rank's own exchange window, system-scope write-through: the peers pull it from there)
    if (dst2 != nullptr) {
        typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_t, s), __builtin_amdgcn_make_buffer_rsrc((void *)(dst2 + i * V), 0, -1, 0x00020000), 0, 0, 17);
    }
}
template <typename T>
__global__ void reduce_slabs_vec_kernel(T *dst, const T *src, int64_t nvec, int nslab, int64_t stride, const int *done) {
    NMFX_DONE_GUARD(done);
    reduce_slabs_vec_body<T>(dst, src, nvec, nslab, stride, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// Same sum for MANY slabs of a small matrix (the tail pieces of the fused Gram: ~128 slabs of k x k).  A thread per
// element would chain `nslab` dependent loads; here 4 slab-lanes per element each add every 4th slab with 8 loads in
// flight, then the 4 partial sums are combined in a fixed order (((l0 + l1) + l2) + l3): deterministic.
template <typename T>
__global__ __launch_bounds__(256) void reduce_many_slabs_kernel(T *dst, const T *src, int64_t count, int nslab, int64_t stride,
                                                                const int *done) {
    NMFX_DONE_GUARD(done);
    __shared__ T sm[4][64];
    const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + e;
    T acc = (T)0;
    if (i < count) {
        int k = sl;
        for (; k + 28 < nslab; k += 32) {
            T v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = src[(int64_t)(k + 4 * q) * stride + i];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += v[q];
        }
        for (; k < nslab; k += 4) acc += src[(int64_t)k * stride + i];
    }
    sm[sl][e] = acc;
    __syncthreads();
    if (sl == 0 && i < count) dst[i] = ((sm[0][e] + sm[1][e]) + sm[2][e]) + sm[3][e];
}

// Two split-K combines in ONE launch (numerator + Gram of the same side: at small shapes every launch is ~5 % of an
// iteration): blocks [0, nb1) serve (dst1, src1, ...), the rest (dst2, src2, ...); same summation order as the kernel above.
template <typename T>
__global__ __launch_bounds__(256) void reduce_pair_kernel(T *dst1, const T *src1, int64_t count1, int nslab1, int64_t stride1, T *dst2,
                                                          const T *src2, int64_t count2, int nslab2, int64_t stride2, unsigned nb1,
                                                          const int *done) {
    NMFX_DONE_GUARD(done);
    __shared__ T sm[4][64];
    const bool first = blockIdx.x < nb1;
    T *dst = first ? dst1 : dst2;
    const T *src = first ? src1 : src2;
    const int64_t count = first ? count1 : count2, stride = first ? stride1 : stride2;
    const int nslab = first ? nslab1 : nslab2;
    const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int64_t i = (int64_t)(first ? blockIdx.x : blockIdx.x - nb1) * 64 + e;
    T acc = (T)0;
    if (i < count) {
        int k = sl;
        for (; k + 28 < nslab; k += 32) {
            T v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = src[(int64_t)(k + 4 * q) * stride + i];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += v[q];
        }
        for (; k < nslab; k += 4) acc += src[(int64_t)k * stride + i];
    }
    sm[sl][e] = acc;
    __syncthreads();
    if (sl == 0 && i < count) dst[i] = ((sm[0][e] + sm[1][e]) + sm[2][e]) + sm[3][e];
}

// A[i + i*ld] += a for i < m  (adddiag!, src/utils.jl:15-24)
// dst (cols x rows, ld ldd) = src' for src (rows x cols, ld lds), both column-major; rows, cols multiples of 64 (padded sizes).
// 64 x 64 tiles through LDS (+1 padding): 256-byte coalesced segments on both sides.  Used ONCE per uploaded X (the second,
// contraction-contiguous image of X for the X*H' product) and once per solve for the start H; never inside an iteration.
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(T *dst, int64_t ldd, const T *src, int64_t lds, int64_t rows, int64_t cols, const int *done) {
    if (done != nullptr && *reinterpret_cast<const volatile int *>(done) != 0) return;
    __shared__ T tile[64][65];
    const int64_t tr = rows / 64;
    const int64_t r0 = (int64_t)(blockIdx.x % tr) * 64, c0 = (int64_t)(blockIdx.x / tr) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) { const int j = ty + 4 * jj; tile[j][tx] = src[(c0 + j) * lds + r0 + tx]; }
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) { const int j = ty + 4 * jj; dst[(r0 + j) * ldd + c0 + tx] = tile[tx][j]; }
}

template <typename T>
__global__ void adddiag_kernel(T *A, int64_t ld, int m, T a, const int *done) {
    NMFX_DONE_GUARD(done);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) A[i + (int64_t)i * ld] += a;
}

// stop_condition statistics for W (src/common.jl:95-99): per column j,
//   dev = sum_i (W-preW)^2, sum = sum_i (W+preW)^2.   grid = (chunks, K); partial[(chunk*K + j)*2 + {0,1}]
template <typename T>
__global__ void col_stats_kernel(const T *Wn, const T *Wo, int64_t rows, int64_t ld, int K, double *partial,
                                 const int *done) {
    NMFX_DONE_GUARD(done);
    __shared__ double sm[8];
    const int j = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const int64_t per = (rows + nchunks - 1) / nchunks;
    const int64_t beg = chunk * per, end = (beg + per < rows) ? beg + per : rows;
    double dev = 0.0, sum = 0.0;
    // four trips' loads in flight, the terms added in trip order (the sums a plain loop forms; a chunk is 1024 rows = four trips at the
    // headline shape, which the plain loop ran as four dependent round trips)
    const int64_t step = blockDim.x;
    const T *wn = Wn + (int64_t)j * ld, *wo = Wo + (int64_t)j * ld;
    for (int64_t i0 = beg + threadIdx.x; i0 < end; i0 += 4 * step) {
        T a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = (i0 + u * step < end) ? i0 + u * step : i0;
            a[u] = wn[i];
            b[u] = wo[i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * step < end) {
                const T d = a[u] - b[u], s = a[u] + b[u];
                dev += (double)(T)(d * d);
                sum += (double)(T)(s * s);
            }
    }
    for (int off = 32; off > 0; off >>= 1) { dev += __shfl_down(dev, off, 64); sum += __shfl_down(sum, off, 64); }
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[w] = dev; sm[4 + w] = sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double d = 0.0, s = 0.0;
        for (int q = 0; q < nw; ++q) { d += sm[q]; s += sm[4 + q]; }
        partial[((int64_t)chunk * K + j) * 2] = d;
        partial[((int64_t)chunk * K + j) * 2 + 1] = s;
    }
}

// stop_condition statistics for H (src/common.jl:100-104): per row j over the local columns.
// grid = (chunks), block = 256 threads, thread <-> row (coalesced across rows).
template <typename T>
__global__ void row_stats_kernel(const T *Hn, const T *Ho, int64_t cols, int64_t ld, int K, double *partial,
                                 const int *done) {
    NMFX_DONE_GUARD(done);
    const int chunk = blockIdx.x, nchunks = gridDim.x;
    const int64_t per = (cols + nchunks - 1) / nchunks;
    const int64_t beg = chunk * per, end = (beg + per < cols) ? beg + per : cols;
    for (int j = threadIdx.x; j < K; j += blockDim.x) {
        double dev = 0.0, sum = 0.0;
        for (int64_t i0 = beg; i0 < end; i0 += 8) {   // eight columns' loads in flight, the terms added in column order
            T a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t i = (i0 + u < end) ? i0 + u : i0;
                a[u] = Hn[j + i * ld];
                b[u] = Ho[j + i * ld];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u < end) {
                    const T d = a[u] - b[u], s = a[u] + b[u];
                    dev += (double)(T)(d * d);
                    sum += (double)(T)(s * s);
                }
        }
        partial[((int64_t)chunk * K + j) * 2] = dev;
        partial[((int64_t)chunk * K + j) * 2 + 1] = sum;
    }
}

// out[e] = sum over chunks of partial[c*stride + e], e < count.  One wave per output: lane l adds chunks
// l, l+64, ... then a fixed shuffle tree -- deterministic, and not a serial chain of `nchunks` dependent loads.
template <typename OutT>
__global__ void finalize_partials_kernel(const double *partial, int nchunks, int stride, int count, OutT *out,
                                         const int *done) {
    NMFX_DONE_GUARD(done);
    const int e = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (e >= count) return;
    const int lane = threadIdx.x & 63;
    double s = 0.0;
    for (int c0 = lane; c0 < nchunks; c0 += 64 * 4) {   // four loads in flight, the adds in chunk order (same sum as a plain loop)
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = (c0 + 64 * u < nchunks) ? c0 + 64 * u : c0;
            v[u] = partial[(int64_t)c * stride + e];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (c0 + 64 * u < nchunks) s += v[u];
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) out[e] = (OutT)s;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also releases global memory, i.e. waits for every outstanding global
// access of the wave (s_waitcnt vmcnt(0): a 1-2 us round trip) -- stores that are never read back inside the kernel, or loads that were
// requested several steps ahead on purpose.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// stop_condition's sums EXACTLY as the reference forms them, second form (round 6; the kernel below is the first, kept for A/B:
// NMFX_STOP_SUMS_V1=1).  The work is 4 k chains of dependent T-precision adds, 2 per factor and component, each as long as the factor is
// tall / wide: a wave issues one such add per 4 cycles whatever the number of active lanes, so a chain of 16384 terms cannot take less
// than 27 us (a lone wave in fact issues one instruction per ~5.3 cycles: 36 us) -- and nothing else may sit in that wave's instruction stream.  Hence: CH = 4 chains per workgroup (k / 4 workgroups per
// factor, both factors in ONE launch: 128 workgroups at k = 256 instead of 16 + 16 one after the other); wave 0 runs the four `dev`
// chains and wave 1 the four `sum` chains, lane <-> chain, and do nothing else -- a 16-byte LDS read per V terms, V dependent adds; waves
// 2 .. 7 are producers: they fetch both factors with 16-byte loads TWO tiles ahead (the barrier between tiles orders LDS only, see
// lds_barrier), form the terms (T)((a - b)^2), (T)((a + b)^2) and stage them [chain][element] in a double-buffered LDS image whose row
// stride (TILE + 16 bytes) puts the four chains' reads on different banks.  Terms past `len` are staged as +0 (sums that are never -0).
// (tile = 2 KiB of terms per chain and kind of sum; 8 KiB tiles -- 131 KB of LDS, 244 registers -- measured the same 95-100 us: the kernel is
// paced by the chain waves' instruction issue, ~2.2 ns per instruction of a lone wave whatever it is (profiles/r04_valu_rate_probe.log),
// at 1.5 instructions per term: the add, a quarter of a 16-byte LDS read and of its s_waitcnt)
constexpr int STOP_SUMS_CH = 4, STOP_SUMS_TILE_BYTES = 2048;
template <typename T> constexpr size_t stop_sums_exact2_lds() { return (size_t)4 * STOP_SUMS_CH * (STOP_SUMS_TILE_BYTES / sizeof(T) + 16 / sizeof(T)) * sizeof(T); }
template <typename T, bool ALONG_ELEM, int CH>
__device__ __forceinline__ void stop_sums_exact2_side(const T *An, const T *Ao, int64_t len, int64_t elem_stride, int64_t chain_stride, int nchains, int c0,
                                                      double *out, T *lds) {
    constexpr int V = 16 / (int)sizeof(T), TILE = STOP_SUMS_TILE_BYTES / (int)sizeof(T), LD = TILE + V, NPROD = 384;
    constexpr int UNITS = CH * TILE / V, ROUNDS = (UNITS + NPROD - 1) / NPROD;
    typedef T vec_t __attribute__((ext_vector_type(V)));
    static_assert(CH % V == 0 || ALONG_ELEM, "a 16-byte load along the chains covers V of them");
    T *sd = lds, *ss = lds + 2 * CH * LD;          // sd[stage][chain][LD], ss likewise
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), ptid = tid - 128;
    const int64_t ntiles = (len + TILE - 1) / TILE;
    vec_t ra[2][ROUNDS], rb[2][ROUNDS];
    bool rin[2][ROUNDS];
    // unit e of a tile: ALONG_ELEM: chain e / (TILE / V), elements V (e % (TILE / V)) ..; else element e / (CH / V), chains V (e % (CH / V)) ..
    auto unit = [&](int e, int &c, int &i) {
        if constexpr (ALONG_ELEM) { c = e / (TILE / V); i = V * (e % (TILE / V)); }
        else { i = e / (CH / V); c = V * (e % (CH / V)); }
    };
    auto fetch = [&](int64_t t, auto SET) {
        constexpr int set = decltype(SET)::value;
#pragma unroll
        for (int u = 0; u < ROUNDS; ++u) {
            const int e = ptid + NPROD * u;
            int c, i;
            unit(e < UNITS ? e : 0, c, i);
            const int64_t ii = t * TILE + i;
            const bool in = e < UNITS && t < ntiles && ii < len;       // (len is a multiple of V: a vector is inside or outside as a whole)
            const int64_t off = in ? (int64_t)(c0 + c) * chain_stride + ii * elem_stride : 0;   // unconditional loads (a clamped address), selected afterwards
            // (the select waits for the data: it happens in stash(), two tiles later -- done here it turned the prefetch into a load-and-wait)
            ra[set][u] = *reinterpret_cast<const vec_t *>(An + off);
            rb[set][u] = *reinterpret_cast<const vec_t *>(Ao + off);
            rin[set][u] = in;
        }
    };
    auto stash = [&](int stage, auto SET) {
        constexpr int set = decltype(SET)::value;
#pragma unroll
        for (int u = 0; u < ROUNDS; ++u) {
            const int e = ptid + NPROD * u;
            if (e >= UNITS) continue;
            int c, i;
            unit(e, c, i);
            vec_t td, ts;
#pragma unroll
            for (int q = 0; q < V; ++q) {
                const T a = rin[set][u] ? ra[set][u][q] : (T)0, b = rin[set][u] ? rb[set][u][q] : (T)0;
                const T d = a - b, sp = a + b;
                td[q] = (T)(d * d);
                ts[q] = (T)(sp * sp);
            }
            if constexpr (ALONG_ELEM) {
                *reinterpret_cast<vec_t *>(sd + (stage * CH + c) * LD + i) = td;
                *reinterpret_cast<vec_t *>(ss + (stage * CH + c) * LD + i) = ts;
            } else {
#pragma unroll
                for (int q = 0; q < V; ++q) { sd[(stage * CH + c + q) * LD + i] = td[q]; ss[(stage * CH + c + q) * LD + i] = ts[q]; }
            }
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    const bool producer = wave >= 2;
    if (producer) {
        fetch(0, S0{});
        fetch(1, S1{});
        stash(0, S0{});
        fetch(2, S0{});
    }
    lds_barrier();
    T acc = (T)0;
    const T *img = (wave == 0) ? sd : ss;
    auto step = [&](int64_t t, auto SET_NEXT) {       // SET_NEXT: the register set that holds tile t + 1
        const int stage = (int)(t & 1);
        if (producer) {
            stash(stage ^ 1, SET_NEXT);
            fetch(t + 3, SET_NEXT);
        } else if (lane < CH) {
            // (the reads of group g + 1 are requested before the adds of group g: read-then-add exposed the ~130 cycles of an LDS round trip
            // once per 8 reads -- as long as the 8 V dependent adds they feed)
            const T *row = img + (stage * CH + lane) * LD;
            constexpr int NG = TILE / (8 * V);
            vec_t x[2][8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[0][u] = *reinterpret_cast<const vec_t *>(row + V * u);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[(g + 1) & 1][u] = *reinterpret_cast<const vec_t *>(row + (g + 1) * 8 * V + V * u);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int q = 0; q < V; ++q) acc = acc + x[g & 1][u][q];
            }
        }
        lds_barrier();
    };
    int64_t t = 0;
    for (; t + 1 < ntiles; t += 2) { step(t, S1{}); step(t + 1, S0{}); }
    if (t < ntiles) step(t, S1{});
    if (wave < 2 && lane < CH && c0 + lane < nchains) out[2 * (c0 + lane) + wave] = (double)acc;
}

// blocks [0, nbw): chains of W (elements contiguous); blocks [nbw, ...): chains of H (chains contiguous), if Hn != nullptr
template <typename T>
__global__ __launch_bounds__(512) void stop_sums_exact2_kernel(const T *Wn, const T *Wo, int64_t P, const T *Hn, const T *Ho, int64_t N, int64_t K, int nchains, int nbw,
                                                               double *wout, double *hout, const int *done) {
    NMFX_DONE_GUARD(done);
    constexpr int CH = STOP_SUMS_CH;
    extern __shared__ __attribute__((aligned(16))) unsigned char stop_sums_smem[];
    T *lds = reinterpret_cast<T *>(stop_sums_smem);
    if ((int)blockIdx.x < nbw) stop_sums_exact2_side<T, true, CH>(Wn, Wo, P, (int64_t)1, P, nchains, (int)blockIdx.x * CH, wout, lds);
    else {
        // 128 / sizeof(T) adjacent chains of H share every 128-byte line, i.e. 8 (Float32) or 4 (Float64) workgroups: give those to ONE XCD
        // (workgroup b runs on XCD b % 8, each with an L2 of its own) -- dealt round-robin, every XCD fetched every line of both factors
        int hb = (int)blockIdx.x - nbw;
        const int nhb = (int)gridDim.x - nbw;
        if ((nhb & 7) == 0) hb = (hb & 7) * (nhb >> 3) + (hb >> 3);
        stop_sums_exact2_side<T, false, CH>(Hn, Ho, N, K, (int64_t)1, nchains, hb * CH, hout, lds);
    }
}

// (first form)
// stop_condition's sums EXACTLY as the reference forms them (src/common.jl:95-104; nmfx_opts.stop_sums = 1): per component j
//     dev_w += (W[i,j] - preW[i,j])^2,  sum_w += (W[i,j] + preW[i,j])^2   for i in index order, accumulated in T
// and the same over row j of H.  One thread per chain pair: the adds are a dependent chain by definition.
// out[2j] = dev, out[2j+1] = sum, stored as the Float64 image of the T-precision totals (check_body rounds back to T: exact).
template <typename T, bool ALONG_ELEM>
__global__ __launch_bounds__(256) void stop_sums_exact_kernel(const T *An, const T *Ao, int64_t len, int64_t elem_stride, int64_t chain_stride, int nchains,
                                                              double *out, const int *done) {
    NMFX_DONE_GUARD(done);
    // A block owns 16 chains (k / 16 blocks: the only parallelism there is, and each block's memory round trip per tile is what paces it).
    // All 256 threads stage tiles of TILE elements x 16 chains of both factors' TERMS through LDS with coalesced 16-byte loads (along
    // whichever of the two indices is contiguous), double-buffered; 16 lanes of wave 0 walk the staged tile, lane c <-> chain c.
    // (The first version let every chain's thread read its own column straight from memory: columns 64 KiB apart alias in every cache
    // level -- 190 cycles per element, 1.3 ms per side at 16384 rows.)
    constexpr int CH = 16, TILE = 1024 / (int)sizeof(T), V = 16 / (int)sizeof(T), PERV = CH * TILE / V / 256;
    typedef T vec_t __attribute__((ext_vector_type(V)));
    constexpr int SZ = (CH * (TILE + 1) > TILE * (CH + 1)) ? CH * (TILE + 1) : TILE * (CH + 1);
    __shared__ T sa[2][SZ], sb[2][SZ];
    const int tid = threadIdx.x, c0 = blockIdx.x * CH;
    constexpr bool along_elem = ALONG_ELEM;        // W (elem_stride == 1): a chain's elements are contiguous; H (chain_stride == 1): the chains are
    // LDS image: [chain][element] (+1) when the loaders' lanes run along the elements, [element][chain] (+1) when they run along the
    // chains -- either way the staging writes and the chain wave's reads (lane <-> chain) spread over the banks (the first layout,
    // [element][chain] for both, put the W side's writes on 8 banks)
    auto at = [&](int c, int i) { return along_elem ? c * (TILE + 1) + i : i * (CH + 1) + c; };
    // 16-byte loads along the contiguous index (the buffers are the padded ones: whole tiles and whole groups of 64 chains exist and
    // are zero beyond the real extent; 4-byte loads made this kernel load-ISSUE-bound: 512 wave-loads per 64-element tile)
    vec_t ra[PERV], rb[PERV];
    auto fetch = [&](int64_t i0) {
#pragma unroll
        for (int u = 0; u < PERV; ++u) {
            const int e = tid + 256 * u;
            const int c = along_elem ? e / (TILE / V) : V * (e % (CH / V)), i = along_elem ? V * (e % (TILE / V)) : e / (CH / V);
            const int64_t off = (int64_t)(c0 + c) * chain_stride + (i0 + i) * elem_stride;
            ra[u] = *reinterpret_cast<const vec_t *>(An + off);
            rb[u] = *reinterpret_cast<const vec_t *>(Ao + off);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < PERV; ++u) {
            const int e = tid + 256 * u;
            const int c = along_elem ? e / (TILE / V) : V * (e % (CH / V)), i = along_elem ? V * (e % (TILE / V)) : e / (CH / V);
            // the TERMS are formed here, by all four waves; the chain wave is left with its two dependent adds per element
#pragma unroll
            for (int q = 0; q < V; ++q) {
                const T d = ra[u][q] - rb[u][q], sp = ra[u][q] + rb[u][q];
                const T td = (T)(d * d), ts = (T)(sp * sp);
                const int ix = along_elem ? at(c, i + q) : at(c + q, i);
                sa[buf][ix] = td;
                sb[buf][ix] = ts;
            }
        }
    };
    T dev = (T)0, sum = (T)0;
    fetch(0);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int64_t i0 = 0; i0 < len; i0 += TILE) {
        const bool more = i0 + TILE < len;
        if (more) fetch(i0 + TILE);
        if (tid < CH) {
            // (elements past `len` were staged as zeros: they add +0 to sums that are never -0.  Eight LDS reads in flight per chain
            // step group: read one by one, every element waited ~100 cycles for its ds_read)
#pragma unroll
            for (int i = 0; i < TILE; i += 8) {
                T x[8], y[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { x[u] = sa[buf][at(tid, i + u)]; y[u] = sb[buf][at(tid, i + u)]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    dev = dev + x[u];
                    sum = sum + y[u];
                }
            }
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    if (tid < CH && c0 + tid < nchains) {
        out[2 * (c0 + tid)] = (double)dev;
        out[2 * (c0 + tid) + 1] = (double)sum;
    }
}

// stop_condition decision (src/common.jl:105-110) + iteration bookkeeping.
// wstat/hstat: [j*2] = dev, [j*2+1] = sum (hstat may be null when update_H is false).
// The reference accumulates in T; the comparison is done in T on the rounded sums.
template <typename T>
__device__ __forceinline__ void check_body(Ctrl *ctrl, const double *wstat, const double *hstat, int k, T tol, long long t, double *dev_out) {
    if (ctrl->done) return;
    __shared__ int first_bad;
    __shared__ T wmax[16];
    if (threadIdx.x == 0) first_bad = k;
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += blockDim.x) {
        const T dw = (T)wstat[2 * j], sw = (T)wstat[2 * j + 1];
        bool b = sqrt(dw) > tol * sqrt(sw);
        if (hstat != nullptr) {
            const T dh = (T)hstat[2 * j], sh = (T)hstat[2 * j + 1];
            b = b || (sqrt(dh) > tol * sqrt(sh));
        }
        if (b) atomicMin(&first_bad, j);
    }
    __syncthreads();
    const int jf = first_bad;
    if (dev_out != nullptr) {
        // devmax of stop_condition (common.jl:93, :106): running maximum of sqrt(max(dev_w/sum_w, dev_h/sum_h)) over the
        // components visited BEFORE the early return, i.e. j <= first failing component.  (update_H = false: the reference's
        // dev_h is exactly 0, so only the W ratio contributes.)
        T m = (T)0;
        const int jend = (jf < k) ? jf : k - 1;
        for (int j = threadIdx.x; j <= jend; j += blockDim.x) {
            T r = (T)wstat[2 * j] / (T)wstat[2 * j + 1];
            if (hstat != nullptr) {
                const T rh = (T)hstat[2 * j] / (T)hstat[2 * j + 1];
                r = (rh > r) ? rh : r;
            }
            const T v = sqrt(r);
            m = (v > m) ? v : m;
        }
        for (int off = 32; off > 0; off >>= 1) {
            const T o = __shfl_down(m, off, 64);
            m = (o > m) ? o : m;
        }
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = (wmax[w] > m) ? wmax[w] : m;
            dev_out[t] = (double)m;
        }
    }
    if (threadIdx.x == 0) {
        ctrl->niters = t;
        if (jf >= k) { ctrl->converged = 1; ctrl->done = 1; }
    }
}
template <typename T>
__global__ void check_kernel(Ctrl *ctrl, const double *wstat, const double *hstat, int k, T tol, long long t, double *dev_out) {
    check_body<T>(ctrl, wstat, hstat, k, tol, t, dev_out);
}

// One-GPU MultUpdate-MSE, nothing tracked: the four launches around the stop rule of an iteration -- stop_condition's H statistics
// finalised from the H update's per-tile partials (finalize_partials_kernel, behind the H update), col_stats_kernel and its
// finalize_partials_kernel behind the W update, check_kernel -- as TWO launches behind the W update (round 6):
//   col_stats_hfin_kernel : every block (chunk, j) column j's sums over its row chunk, as col_stats_kernel (same bits); blocks
//                           0 .. (2K + 3) / 4 - 1 also four of the 2K H statistics each, a wave per output as finalize_partials_kernel
//                           (same bits) -- work that fits beside 4096 blocks of column sums for free
//   wfin_check_kernel     : ONE block: the W partials added up in chunk order (a thread per output, every load of a thread in flight
//                           at once), then check_body
// (Measured first as ONE launch whose last-arriving block did the second half: 4096 blocks taking an agent-scope ticket on one
// address cost 80 us -- ~20 ns per contended atomic -- against the 23.5 us of the four launches.)
template <typename T>
__global__ __launch_bounds__(256) void col_stats_hfin_kernel(const T *Wn, const T *Wo, int64_t rows, int64_t ld, int K, double *wpart, const double *hpart,
                                                             int hchunks, double *hstat, const int *done) {
    NMFX_DONE_GUARD(done);
    __shared__ double sm[8];
    const int j = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const unsigned flat = blockIdx.y * gridDim.x + blockIdx.x;
    {
        const int64_t per = (rows + nchunks - 1) / nchunks;
        const int64_t beg = chunk * per, end = (beg + per < rows) ? beg + per : rows;
        double dev = 0.0, sum = 0.0;
        const int64_t step = blockDim.x;
        const T *wn = Wn + (int64_t)j * ld, *wo = Wo + (int64_t)j * ld;
        for (int64_t i0 = beg + threadIdx.x; i0 < end; i0 += 4 * step) {
            T a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = (i0 + u * step < end) ? i0 + u * step : i0;
                a[u] = wn[i];
                b[u] = wo[i];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * step < end) {
                    const T d = a[u] - b[u], s = a[u] + b[u];
                    dev += (double)(T)(d * d);
                    sum += (double)(T)(s * s);
                }
        }
        for (int off = 32; off > 0; off >>= 1) { dev += __shfl_down(dev, off, 64); sum += __shfl_down(sum, off, 64); }
        const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
        if ((threadIdx.x & 63) == 0) { sm[w] = dev; sm[4 + w] = sum; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double d = 0.0, s = 0.0;
            for (int q = 0; q < nw; ++q) { d += sm[q]; s += sm[4 + q]; }
            wpart[((int64_t)chunk * K + j) * 2] = d;
            wpart[((int64_t)chunk * K + j) * 2 + 1] = s;
        }
    }
    if (hpart != nullptr) {
        const int e = (int)flat * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);
        if (e < 2 * K) {
            const int lane = threadIdx.x & 63;
            double s = 0.0;
            for (int c0 = lane; c0 < hchunks; c0 += 64 * 4) {
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = (c0 + 64 * u < hchunks) ? c0 + 64 * u : c0;
                    v[u] = hpart[(int64_t)c * 2 * K + e];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (c0 + 64 * u < hchunks) s += v[u];
            }
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (lane == 0) hstat[e] = s;
        }
    }
}
template <typename T>
__global__ __launch_bounds__(1024) void wfin_check_kernel(const double *wpart, int nchunks, int K, double *wstat, const double *hstat, Ctrl *ctrl, int k, T tol,
                                                         long long t, const int *done) {
    NMFX_DONE_GUARD(done);
    for (int e = threadIdx.x; e < 2 * K; e += blockDim.x) {
        double s = 0.0;
        for (int c0 = 0; c0 < nchunks; c0 += 32) {
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = wpart[(int64_t)((c0 + u < nchunks) ? c0 + u : nchunks - 1) * 2 * K + e];
#pragma unroll
            for (int u = 0; u < 32; ++u)
                if (c0 + u < nchunks) s += v[u];
        }
        wstat[e] = s;
    }
    __syncthreads();
    check_body<T>(ctrl, wstat, hstat, k, tol, t, nullptr);
}

// objective finalisation (evaluate_objv): s = sum of block partials (ascending);
//   mode 0: out = T(0.5*s + extra)   (src/multupd.jl:81, src/projals.jl:65-74, src/alspgrad.jl:398)
//   mode 1: out = T(s)               (gkldiv, src/multupd.jl:148)
// `extra` (nullable) holds already-scaled regularisation terms (projals).
template <typename T>
__global__ void finish_objective_kernel(const double *partial, int n, int mode, const double *extra, int nextra,
                                        double *out, const int *done) {
    NMFX_DONE_GUARD(done);
    __shared__ double sm[4];
    double v = 0.0;
    // fixed order: thread-strided then lane tree then wave order
    for (int i = threadIdx.x; i < n; i += blockDim.x) v += partial[i];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sm[w];
        double r = (mode == 0) ? 0.5 * s : s;
        for (int e = 0; e < nextra; ++e) r += extra[e];
        *out = (double)(T)r;   // Result{T} conversion (src/common.jl:33)
    }
}

// generic strided "sum over one axis" used for multdiv's sW = sum(W, dims=1) and sH = sum(H, dims=2)
// (src/multupd.jl:176,188) -- same two-stage shape as the stats kernels, one value per component.
template <typename T>
__global__ void col_sum_kernel(const T *W, int64_t rows, int64_t ld, int K, double *partial, const int *done) {
    NMFX_DONE_GUARD(done);
    __shared__ double sm[4];
    const int j = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const int64_t per = (rows + nchunks - 1) / nchunks;
    const int64_t beg = chunk * per, end = (beg + per < rows) ? beg + per : rows;
    double s = 0.0;
    for (int64_t i = beg + threadIdx.x; i < end; i += blockDim.x) s += (double)W[i + (int64_t)j * ld];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int q = 0; q < (int)(blockDim.x >> 6); ++q) t += sm[q];
        partial[(int64_t)chunk * K + j] = t;
    }
}

template <typename T>
__global__ void row_sum_kernel(const T *H, int64_t cols, int64_t ld, int K, double *partial, const int *done) {
    NMFX_DONE_GUARD(done);
    const int chunk = blockIdx.x, nchunks = gridDim.x;
    const int64_t per = (cols + nchunks - 1) / nchunks;
    const int64_t beg = chunk * per, end = (beg + per < cols) ? beg + per : cols;
    for (int j = threadIdx.x; j < K; j += blockDim.x) {
        double s = 0.0;
        for (int64_t i = beg; i < end; ++i) s += (double)H[j + i * ld];
        partial[(int64_t)chunk * K + j] = s;
    }
}

// multdiv scalings (src/multupd.jl:177-179, 189-191) over the LOGICAL rows x cols extent (padding is never touched:
// with lambda == 0 a padded component would give 0 * (0/0) = NaN there):
//   along_contig = 1 (H, k x n): out[i + j*ld] = in * num / (s[i] + lambda)   (s indexed by the contiguous index)
//   along_contig = 0 (W, p x k): out[i + j*ld] = in * num / (s[j] + lambda)
// flat 1-D grid-stride index: no gridDim.y limit on the number of columns of a shard.
template <typename T>
__global__ void div_update_kernel(T *out, const T *in, const T *num, const T *s, int64_t rows, int64_t cols,
                                  int64_t ld, T lambda, int along_contig, const int *done) {
    NMFX_DONE_GUARD(done);
    const int64_t total = rows * cols;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e % rows, j = e / rows;
        const int64_t o = i + j * ld;
        const T d = (along_contig ? s[i] : s[j]) + lambda;
        out[o] = in[o] * (num[o] / d);
    }
}

// sum of squares in Float64 -> partial per block (projals objective ||W||^2, ||H||^2: src/projals.jl:67-72)
template <typename T>
__global__ void sumsq_kernel(const T *A, int64_t count, double *partial, const int *done) {
    NMFX_DONE_GUARD(done);
    __shared__ double sm[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const T a = A[i];
        s += (double)(T)(a * a);
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int q = 0; q < (int)(blockDim.x >> 6); ++q) t += sm[q];
        partial[blockIdx.x] = t;
    }
}

// extra[slot] = T(0.5*lambda) * T(sum partial)   (projals regulariser term, in T like the reference)
template <typename T>
__global__ void finish_sumsq_kernel(const double *partial, int n, T half_lambda, double *extra, int slot, const int *done) {
    NMFX_DONE_GUARD(done);
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += partial[i];
    const T nrm = sqrt((T)s);           // abs2(norm(W)) with norm in T
    extra[slot] = (double)(T)(half_lambda * (nrm * nrm));
}

// ---------------------------------------------------------------------------------------------------------------------
// Row-sharded W side of the multi-GPU path: rank g owns rows [g*Pc, (g+1)*Pc) of the W update.
// A "blocked" P x K matrix is G consecutive Pc x K column-major pieces (piece g = rows of rank g, ld Pc): the layout in
// which a reduce-scatter / all-gather chunk is contiguous.
// ---------------------------------------------------------------------------------------------------------------------

// split-K combine of the X*H' slabs written STRAIGHT into the blocked reduce-scatter send buffer, for the rows [i0, i0 + R)
// (all P rows, or one row super-chunk of the pipelined exchange):
//   dst[(blk*K + a)*Pc + il] = sum_s src[s*stride + i + a*P],  i = blk*Pc + il   (s ascending, like reduce_slabs_kernel)
template <typename T>
__global__ void reduce_slabs_blocked_kernel(T *dst, const T *src, int64_t P, int64_t K, int64_t Pc, int nslab, int64_t stride,
                                            int64_t i0, int64_t R, const int *done) {
    NMFX_DONE_GUARD(done);
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= R * K) return;
    const int64_t i = i0 + e % R, a = e / R, blk = i / Pc, il = i % Pc, o = i + a * P;
    T s = src[o];
    for (int k = 1; k < nslab; ++k) s += src[(int64_t)k * stride + o];
    dst[(blk * K + a) * Pc + il] = s;
}

// rows [row0, row0 + Pc) of a column-major P x K matrix (ld P)  <->  a contiguous Pc x K piece (ld Pc)
template <typename T>
__global__ void rows_to_piece_kernel(T *piece, const T *full, int64_t P, int64_t K, int64_t Pc, int64_t row0, const int *done) {
    NMFX_DONE_GUARD(done);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < Pc * K; e += (int64_t)gridDim.x * blockDim.x)
        piece[e] = full[row0 + e % Pc + (e / Pc) * P];
}
template <typename T>
__global__ void piece_to_rows_kernel(T *full, const T *piece, int64_t P, int64_t K, int64_t Pc, int64_t row0, const int *done) {
    NMFX_DONE_GUARD(done);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < Pc * K; e += (int64_t)gridDim.x * blockDim.x)
        full[row0 + e % Pc + (e / Pc) * P] = piece[e];
}
// all-gather receive buffer (G chunks of `chunk_bytes`: [Pc x K piece | ntail doubles]) -> rows [rowbase + g*Pc, ... + Pc) of the
// full matrix (ld P), g = 0..G-1; the ranks' tail vectors (per-column partial sums of the stop statistics,
// src/common.jl:95-99) are added in rank order -> tail_sum (accumulate != 0: added to what is there -- the later row
// super-chunks of the pipelined exchange)
template <typename T>
__global__ void gathered_to_full_kernel(T *full, const unsigned char *recv, int G, size_t chunk_bytes, int64_t P, int64_t K,
                                        int64_t Pc, int64_t rowbase, int ntail, double *tail_sum, int accumulate, const int *done) {
    NMFX_DONE_GUARD(done);
    const int64_t total = (int64_t)G * Pc * K;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = e / (Pc * K), r = e % (Pc * K);
        const T *piece = reinterpret_cast<const T *>(recv + (size_t)g * chunk_bytes);
        full[rowbase + g * Pc + r % Pc + (r / Pc) * P] = piece[r];
    }
    if (blockIdx.x == 0 && tail_sum != nullptr)
        for (int j = threadIdx.x; j < ntail; j += blockDim.x) {
            double s = accumulate ? tail_sum[j] : 0.0;
            for (int g = 0; g < G; ++g)
                s += reinterpret_cast<const double *>(recv + (size_t)g * chunk_bytes + (size_t)Pc * K * sizeof(T))[j];
            tail_sum[j] = s;
        }
}

// The ranks' (or chunks') stop_condition partials added in chunk order, then the stop rule -- by NB = gridDim blocks (round 6; one block
// before): block b owns the outputs e in [b * 2K / NB, (b + 1) * 2K / NB), one thread per output, 32 of the output's chunk values
// requested before the first add (the adds stay in chunk order: the bits of the one-block form, whose 2K sums of 64 values each were
// four dependent round trips per thread, 10.5 us at the 8-rank shard shape), and the LAST block to arrive (a ticket: NB agent-scope
// atomics) runs check_body on the complete wstat.  `ld(c, e)` = partial of chunk c, output e.
template <typename T, typename LD>
__device__ __forceinline__ void stats_sum_check(LD ld, int nchunks, int K, double *wstat, unsigned *ticket, unsigned b, unsigned nb, Ctrl *ctrl, const double *hstat,
                                                int k, T tol, long long t, int do_check) {
    const int per = (2 * K + (int)nb - 1) / (int)nb;
    const int e0 = (int)b * per, e1 = (e0 + per < 2 * K) ? e0 + per : 2 * K;
    for (int e = e0 + (int)threadIdx.x; e < e1; e += (int)blockDim.x) {
        double s = 0.0;
        for (int c0 = 0; c0 < nchunks; c0 += 32) {
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = ld((c0 + u < nchunks) ? c0 + u : nchunks - 1, e);
#pragma unroll
            for (int u = 0; u < 32; ++u)
                if (c0 + u < nchunks) s += v[u];
        }
        wstat[e] = s;
    }
    __shared__ int last_sm;
    __syncthreads();
    if (threadIdx.x == 0) {
        int last = 1;
        if (nb > 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (old % nb) == nb - 1;    // (the counter only ever grows: every launch adds exactly nb, or nothing behind the stop)
            if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        last_sm = last;
    }
    __syncthreads();
    if (last_sm && do_check) check_body<T>(ctrl, wstat, hstat, k, tol, t, nullptr);
}
// The stop rule of the PREVIOUS iteration, riding in this iteration's combine launch (round 6): the row-sharded fused
// step no longer ends in a one-block launch that adds the ranks' statistics and decides -- the next iteration's w_side_combine_kernel
// carries nb extra blocks that do it (stats_sum_check on the statistics tails of the blocked W buffer).  Everything the next iteration
// enqueued in front of its combine launch wrote ping-pong partners or scratch only, so a stop found here leaves iteration t's factors
// intact.  nb == 0: nothing deferred.
template <typename T> struct DeferChk {
    const double *tails;     // statistics tail of rank 0's chunk in the blocked W buffer of iteration t
    int nchunks, grp;        // chunks in total, chunks per rank
    int64_t grp_stride;      // doubles between two ranks' tails
    double *wstat;
    Ctrl *ctrl;
    const double *hstat;     // iteration t's H statistics (all-reduced), or nullptr
    unsigned *ticket;
    int K, k;
    T tol;
    long long t;
    unsigned nb;
};
// Everything between the X_g H_g' launch and the exchange of the row-sharded MultUpdate-MSE step in ONE launch (three launches of
// 6-17 us each at the 8-rank shard shape of the headline problem, where an iteration is 0.4 ms):
//   blocks [0, nb1)         : split-K combine of the numerator slabs into the blocked send buffer (reduce_slabs_blocked_kernel)
//   blocks [nb1, nb1 + nb2) : the fused Gram's tail pieces summed (reduce_many_slabs_kernel: 4 slab-lanes per element, fixed order)
//   the rest                : stop_condition's H statistics finalised from the update GEMM's per-tile partials
//                             (finalize_partials_kernel: a wave per output)
// Destinations of the three outputs on the peer-to-peer transport (peer.hpp): piece g of the numerator goes to num[g] (rank g's receive
// slot), the Gram and the statistics to EVERY rank's slot (n > 0; system-scope write-through stores).  n == 0: the local buffers.
template <typename T> struct CombineDst {
    T *num[16];
    T *gram[16];
    double *hstat[16];
    int n;
};
template <typename T> __device__ __forceinline__ void peer_store(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
template <typename T>
__global__ __launch_bounds__(256) void w_side_combine_kernel(T *dst_blk, const T *slabs, int64_t P, int64_t K, int64_t Pc, int nslab, int64_t stride,
                                                             T *gram, const T *gsrc, int64_t gcount, int gslabs, int64_t gstride,
                                                             const double *hpart, int hchunks, int hcount, double *hstat, unsigned nb1,
                                                             unsigned nb2, const int *done, CombineDst<T> pd, DeferChk<T> dc) {
    NMFX_DONE_GUARD(done);
    __shared__ T sm[4][64];
    if (blockIdx.x >= gridDim.x - dc.nb) {   // the deferred stop rule of the iteration before (see DeferChk)
        auto ld = [&](int c, int e) { return dc.tails[(int64_t)(c / dc.grp) * dc.grp_stride + (int64_t)(c % dc.grp) * 2 * dc.K + e]; };
        stats_sum_check<T>(ld, dc.nchunks, dc.K, dc.wstat, dc.ticket, blockIdx.x - (gridDim.x - dc.nb), dc.nb, dc.ctrl, dc.hstat, dc.k, dc.tol, dc.t, 1);
        return;
    }
    if (blockIdx.x < nb1) {
        // 256 consecutive rows of one column per block (P is a multiple of 256): 32-bit index arithmetic, no 64-bit division
        const unsigned rb = (unsigned)(P / 256), a = blockIdx.x / rb, i = (blockIdx.x % rb) * 256u + threadIdx.x;
        const unsigned blk = i / (unsigned)Pc, il = i - blk * (unsigned)Pc;
        const int64_t o = (int64_t)i + (int64_t)a * P;
        T s = slabs[o];
        for (int q = 1; q < nslab; ++q) s += slabs[(int64_t)q * stride + o];
        if (pd.n > 0) peer_store(pd.num[blk] + (int64_t)a * Pc + il, s);
        else dst_blk[((int64_t)blk * K + a) * Pc + il] = s;
    } else if (blockIdx.x < nb1 + nb2) {
        const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
        const int64_t i = (int64_t)(blockIdx.x - nb1) * 64 + e;
        T acc = (T)0;
        if (i < gcount) {
            int q = sl;
            for (; q + 28 < gslabs; q += 32) {
                T v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = gsrc[(int64_t)(q + 4 * u) * gstride + i];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
            for (; q < gslabs; q += 4) acc += gsrc[(int64_t)q * gstride + i];
        }
        sm[sl][e] = acc;
        __syncthreads();
        if (sl == 0 && i < gcount) {
            const T v = ((sm[0][e] + sm[1][e]) + sm[2][e]) + sm[3][e];
            if (pd.n > 0) { for (int q = 0; q < pd.n; ++q) peer_store(pd.gram[q] + i, v); }
            else gram[i] = v;
        }
    } else {
        const int e = (int)(blockIdx.x - nb1 - nb2) * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);
        if (e >= hcount) return;
        const int lane = threadIdx.x & 63;
        double s = 0.0;
        for (int c = lane; c < hchunks; c += 64) s += hpart[(int64_t)c * hcount + e];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) {
            if (pd.n > 0) { for (int q = 0; q < pd.n; ++q) peer_store(pd.hstat[q] + e, s); }
            else hstat[e] = s;
        }
    }
}

// The read side of the row-sharded W step's first exchange on the peer-to-peer transport, ONE launch: this rank's rows of the numerator,
// the Gram H H' and the H statistics, each the sum of the n ranks' contributions IN RANK ORDER (slot q at base + q * slot_bytes; the order
// of the in-process group's reduction kernels, so both transports give the same bits).  blocks [0, nb1): numerator, [nb1, nb1 + nb2):
// Gram, the rest: statistics (nstat may be 0).
// dst[i] = src_0[i] + src_1[i] + ... (slot q at src + q * slot_bytes), rank order, V elements per access; all n <= 16 loads of an element are
// requested before the first add (the slots are UNCACHED window memory: as `s += load` in a loop over a run-time count the seven loads of
// an 8-rank sum were a chain of dependent fabric round trips -- most of the 16.7 us this launch took at the 8-rank shard shape)
template <typename T, int V, int NMAX>   // NMAX = n rounded up to 2, 4, 8, 16 (a surplus load re-reads the last slot and is not added)
__device__ __forceinline__ void peer_sum_slots_n(T *dst, const unsigned char *src, size_t count, size_t slot_bytes, int n, size_t first, size_t stride) {
    typedef T vec_t __attribute__((ext_vector_type(V)));
    auto ld = [](const unsigned char *p) { if constexpr (V == 1) return *reinterpret_cast<const T *>(p); else return *reinterpret_cast<const vec_t *>(p); };
    for (size_t i = first; i < count / V; i += stride) {
        const unsigned char *b = src + i * V * sizeof(T);
        decltype(ld(b)) v[NMAX];
#pragma unroll
        for (int u = 0; u < NMAX; ++u) v[u] = ld(b + (size_t)(u < n ? u : n - 1) * slot_bytes);
        auto s = v[0];
#pragma unroll
        for (int u = 1; u < NMAX; ++u)
            if (u < n) s += v[u];
        for (int q = NMAX; q < n; ++q) s += ld(b + (size_t)q * slot_bytes);
        if constexpr (V == 1) dst[i] = s;
        else *reinterpret_cast<vec_t *>(dst + i * V) = s;
    }
}
template <typename T, int V>
__device__ __forceinline__ void peer_sum_slots(T *dst, const unsigned char *src, size_t count, size_t slot_bytes, int n, size_t first, size_t stride) {
    if (n <= 2) peer_sum_slots_n<T, V, 2>(dst, src, count, slot_bytes, n, first, stride);
    else if (n <= 4) peer_sum_slots_n<T, V, 4>(dst, src, count, slot_bytes, n, first, stride);
    else if (n <= 8) peer_sum_slots_n<T, V, 8>(dst, src, count, slot_bytes, n, first, stride);
    else peer_sum_slots_n<T, V, 16>(dst, src, count, slot_bytes, n, first, stride);
}
template <typename T>
__device__ __forceinline__ void peer_sum_slots_any(T *dst, const unsigned char *src, size_t count, size_t slot_bytes, int n, size_t first, size_t stride) {
    constexpr int V = 16 / (int)sizeof(T);
    const bool vec = count % V == 0 && slot_bytes % 16 == 0 && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0;
    if (vec) peer_sum_slots<T, V>(dst, src, count, slot_bytes, n, first, stride);
    else peer_sum_slots<T, 1>(dst, src, count, slot_bytes, n, first, stride);
}
template <typename T>
__global__ __launch_bounds__(256) void peer_sum3_kernel(T *num, const unsigned char *num_src, size_t nnum, T *gram, const unsigned char *gram_src, size_t ngram,
                                                        double *stat, const unsigned char *stat_src, size_t nstat, size_t slot_bytes, int n, unsigned nb1,
                                                        unsigned nb2, const int *done) {
    NMFX_DONE_GUARD(done);
    if (blockIdx.x < nb1) {
        peer_sum_slots_any<T>(num, num_src, nnum, slot_bytes, n, (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)nb1 * blockDim.x);
    } else if (blockIdx.x < nb1 + nb2) {
        peer_sum_slots_any<T>(gram, gram_src, ngram, slot_bytes, n, (size_t)(blockIdx.x - nb1) * blockDim.x + threadIdx.x, (size_t)nb2 * blockDim.x);
    } else {
        const unsigned nb3 = gridDim.x - nb1 - nb2;
        peer_sum_slots<double, 1>(stat, stat_src, nstat, slot_bytes, n, (size_t)(blockIdx.x - nb1 - nb2) * blockDim.x + threadIdx.x, (size_t)nb3 * blockDim.x);
    }
}

// The tail of the row-sharded W side: the all-gather's receive buffer (G pieces of Pc x K, ld Pc) is unpacked into the full W
// (ld P) and stop_condition's column sums (src/common.jl:95-99) are taken over ALL rows against the old W on the way -- every rank
// computes the same sums from the same bits, so no statistics travel with the all-gather.  grid = (G * cpp, K): cpp chunks per
// piece, a block stays inside one rank's piece; partial[(chunk*K + j)*2 + {0,1}].  Followed by stats_check_kernel (one block): the
// per-chunk partials added in chunk order and, unless the caller still has an objective evaluation to enqueue in front of it
// (do_check = 0), the stop rule itself.  Two launches instead of col_stats + finalize + rows_to_piece + gathered_to_full + check.
// (One launch with a last-block ticket was measured: 4096 device-scope atomics on one address cost 130 us on this chip.)
// Generalised for the blocked residency of W (solver_impl.hpp: multmse_w_rows_fused): Wfull == nullptr takes the sums only (no
// unpacking); the old factor's row block g starts at Wold + g * old_blk with leading dimension ldo (standard layout: old_blk = Pc,
// ldo = P; blocked: the chunk stride in elements, ldo = Pc); g_first = the first row block of the grid (own-rows statistics: the
// rank's block only, grid.x = cpp) -- the per-chunk arithmetic, hence every partial's bits, is the same in all forms.
template <typename T>
__device__ __forceinline__ void gather_stats_body(int chunk, int j, T *Wfull, const T *Wold, const unsigned char *recv, size_t chunk_bytes, int64_t P,
                                                  int64_t Pc, int cpp, int K, double *partial, int64_t old_blk, int64_t ldo, int g_first, double *partial2 = nullptr) {
    __shared__ double sm[8];
    const int g = g_first + chunk / cpp, ci = chunk % cpp;
    const int64_t per = (Pc + cpp - 1) / cpp;
    const int64_t beg = ci * per, end = (beg + per < Pc) ? beg + per : Pc;
    if (old_blk < 0) { old_blk = Pc; ldo = P; }
    const T *piece = reinterpret_cast<const T *>(recv + (size_t)g * chunk_bytes) + (int64_t)j * Pc;
    const T *oldc = Wold + (int64_t)g * old_blk + (int64_t)j * ldo;
    T *newc = Wfull + (int64_t)g * Pc + (int64_t)j * P;
    const bool copy = Wfull != nullptr;
    double dev = 0.0, sum = 0.0;
    // 16-byte accesses (Pc and P are multiples of 128 rows, the chunk bounds multiples of 4 whenever Pc / cpp is): a thread's
    // VEC consecutive rows per trip
    constexpr int VEC = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    if (((beg | end) & (VEC - 1)) == 0) {
        for (int64_t il = beg + (int64_t)threadIdx.x * VEC; il < end; il += (int64_t)blockDim.x * VEC) {
            const vec_t a = *reinterpret_cast<const vec_t *>(piece + il);
            const vec_t b = *reinterpret_cast<const vec_t *>(oldc + il);
            if (copy) *reinterpret_cast<vec_t *>(newc + il) = a;
#pragma unroll
            for (int u = 0; u < VEC; ++u) {
                const T d = a[u] - b[u], s = a[u] + b[u];
                dev += (double)(T)(d * d);
                sum += (double)(T)(s * s);
            }
        }
    } else {
        for (int64_t il = beg + threadIdx.x; il < end; il += blockDim.x) {
            const T a = piece[il];
            const T b = oldc[il];
            if (copy) newc[il] = a;
            const T d = a - b, s = a + b;
            dev += (double)(T)(d * d);
            sum += (double)(T)(s * s);
        }
    }
    for (int off = 32; off > 0; off >>= 1) { dev += __shfl_down(dev, off, 64); sum += __shfl_down(sum, off, 64); }
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[w] = dev; sm[4 + w] = sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double d = 0.0, s = 0.0;
        for (int q = 0; q < nw; ++q) { d += sm[q]; s += sm[4 + q]; }
        partial[((int64_t)chunk * K + j) * 2] = d;
        partial[((int64_t)chunk * K + j) * 2 + 1] = s;
        if (partial2 != nullptr) {   // ... and into the rank's own exchange window (pulled by the peers)
            __hip_atomic_store(partial2 + ((int64_t)chunk * K + j) * 2, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(partial2 + ((int64_t)chunk * K + j) * 2 + 1, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void gather_stats_kernel(T *Wfull, const T *Wold, const unsigned char *recv, size_t chunk_bytes, int64_t P,
                                                           int64_t Pc, int cpp, int K, double *partial, const int *done, int64_t old_blk = -1,
                                                           int64_t ldo = -1, int g_first = 0) {
    NMFX_DONE_GUARD(done);
    gather_stats_body<T>((int)blockIdx.x, (int)blockIdx.y, Wfull, Wold, recv, chunk_bytes, P, Pc, cpp, K, partial, old_blk, ldo, g_first);
}
// Behind the rank's W update in the blocked-residency step: its two independent small passes in ONE launch -- blocks [0, nbs): the
// stop_condition sums of the rank's own rows (gather_stats_body, block b <-> chunk b % nchunk, component b / nchunk); the rest: the
// split-K combine of the own-rows Gram (reduce_slabs_vec_body).
template <typename T>
__global__ __launch_bounds__(256) void rows_tail_kernel(const T *Wold, const unsigned char *recv, size_t chunk_bytes, int64_t P, int64_t Pc, int cpp, int K,
                                                        double *partial, int64_t old_blk, int64_t ldo, int g_first, unsigned nbs, T *gdst, const T *gsrc,
                                                        int64_t gnvec, int gslabs, int64_t gstride, const int *done, double *partial2 = nullptr, T *gdst2 = nullptr) {
    NMFX_DONE_GUARD(done);
    if (blockIdx.x < nbs) {
        gather_stats_body<T>((int)(blockIdx.x % (unsigned)cpp), (int)(blockIdx.x / (unsigned)cpp), (T *)nullptr, Wold, recv, chunk_bytes, P, Pc, cpp, K, partial, old_blk,
                             ldo, g_first, partial2);
    } else {
        reduce_slabs_vec_body<T>(gdst, gsrc, gnvec, gslabs, gstride, (int64_t)(blockIdx.x - nbs) * blockDim.x + threadIdx.x, gdst2);
    }
}
template <typename T>
// grp > 0: the partials arrive in groups of grp chunks, group q at partial + q * grp_stride doubles (the ranks' statistics tails
// behind their row blocks in the blocked W buffer); summed in the same chunk order either way.
__global__ __launch_bounds__(256) void stats_check_kernel(const double *partial, int nchunks, int K, double *wstat, Ctrl *ctrl, const double *hstat,
                                                          int k, T tol, long long t, int do_check, const int *done, int grp = 0, int64_t grp_stride = 0,
                                                          unsigned *ticket = nullptr) {
    NMFX_DONE_GUARD(done);
    auto ld = [&](int c, int e) {
        return (grp > 0) ? partial[(int64_t)(c / grp) * grp_stride + (int64_t)(c % grp) * 2 * K + e] : partial[(int64_t)c * 2 * K + e];
    };
    stats_sum_check<T>(ld, nchunks, K, wstat, ticket, blockIdx.x, gridDim.x, ctrl, hstat, k, tol, t, do_check);
}

// The second exchange of the row-sharded MultUpdate-MSE step on the peer-to-peer transport as a PULL, ONE launch behind the flag wait
// (solver_impl.hpp: multmse_w_rows_fused_peer).  Every rank has left its chunk [ Pc x K new rows | cpp x 2K statistics partials ] and
// its own-rows Gram in ITS OWN window (the producing kernels store them there); here
//   blocks [0, n * nbc)          : the peers' chunks copied out of THEIR windows (16-byte system-scope loads over xGMI, every link at
//                                  once) into this rank's blocked W buffer -- the layout the next W'X contracts over in place; the own
//                                  chunk is already there (cached copy written by the update's epilogue)
//   blocks [n * nbc, .. + nbg)   : W'W = the ranks' own-rows Grams added in rank order (peer_sum_slots' order)
//   the last STAT_BLOCKS blocks  : the ranks' statistics partials added in chunk order straight out of the windows and, with do_check,
//                                  the stop rule -- stats_check_kernel's arithmetic (stats_sum_check: same bits), without its launch
// No push launch and no unpack launch: 2 MB per link are READ once, where the push form wrote 16 MB of uncached stores and read them
// back.  The copy / Gram blocks carry no `done` guard when the stop rule runs in this very launch (the last block may raise the flag
// while they are still being dispatched); behind a stop they re-deliver what the windows still hold, which nothing reads.
struct PullSrc {
    const unsigned char *chunk[16];   // rank q's chunk as mapped here
    const unsigned char *gram[16];    // rank q's own-rows Gram as mapped here
};
template <typename T>
__global__ __launch_bounds__(256) void peer_pull_kernel(PullSrc ps, int rank, int n, unsigned char *dst, size_t chunk_bytes, unsigned nbc, T *gram, size_t ngram,
                                                        unsigned nbg, size_t tail_off, int cpp, int K, double *wstat, Ctrl *ctrl, const double *hstat, int k,
                                                        T tol, long long t, int do_check, const int *done_copy, const int *done, unsigned *ticket) {
    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
    // system-scope 16-byte load at base + off: descriptor from the WAVE-UNIFORM base, the lane's part as the 32-bit offset operand (a
    // descriptor built from a per-lane pointer is wrapped in a waterfall loop: 64 serial one-lane loads per instruction; peer.hpp)
    auto sld = [](const unsigned char *base_uniform, uint32_t off) {
        return __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc((void *)base_uniform, 0, -1, 0x00020000), (int)off, 0, 17);
    };
    const unsigned b = blockIdx.x;
    if (b < (unsigned)n * nbc) {
        NMFX_DONE_GUARD(done_copy);
        const int q = (int)(b / nbc);
        if (q == rank) return;
        const unsigned char *s = ps.chunk[q];
        v4u_t *d = reinterpret_cast<v4u_t *>(dst + (size_t)q * chunk_bytes);
        const uint32_t nv = (uint32_t)(chunk_bytes / 16), stride = nbc * blockDim.x;   // (chunk_bytes < 4 GiB: checked by the host)
        uint32_t i = (b % nbc) * blockDim.x + threadIdx.x;
        for (; i + 3 * stride < nv; i += 4 * stride) {   // four loads in flight per lane
            const v4u_t v0 = sld(s, i * 16u), v1 = sld(s, (i + stride) * 16u), v2 = sld(s, (i + 2 * stride) * 16u), v3 = sld(s, (i + 3 * stride) * 16u);
            d[i] = v0;
            d[i + stride] = v1;
            d[i + 2 * stride] = v2;
            d[i + 3 * stride] = v3;
        }
        for (; i < nv; i += stride) d[i] = sld(s, i * 16u);
    } else if (b < (unsigned)n * nbc + nbg) {
        NMFX_DONE_GUARD(done_copy);
        constexpr int V = 16 / (int)sizeof(T);
        typedef T vec_t __attribute__((ext_vector_type(V)));
        const uint32_t nv = (uint32_t)(ngram / V);
        for (uint32_t i = (b - (unsigned)n * nbc) * blockDim.x + threadIdx.x; i < nv; i += nbg * blockDim.x) {
            vec_t v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = __builtin_bit_cast(vec_t, sld(ps.gram[u < n ? u : n - 1], i * 16u));
            vec_t s = v[0];
#pragma unroll
            for (int u = 1; u < 16; ++u)
                if (u < n) s += v[u];
            *reinterpret_cast<vec_t *>(gram + i * V) = s;
        }
    } else {
        NMFX_DONE_GUARD(done);
        auto ld = [&](int c, int e) {
            return __hip_atomic_load(reinterpret_cast<const double *>(ps.chunk[c / cpp] + tail_off) + (int64_t)(c % cpp) * 2 * K + e, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_SYSTEM);
        };
        stats_sum_check<T>(ld, n * cpp, K, wstat, ticket, b - ((unsigned)n * nbc + nbg), gridDim.x - ((unsigned)n * nbc + nbg), ctrl, hstat, k, tol, t, do_check);
    }
}

// dst[c + r*ldd] = sum_p src[p*stride + c + r*cols] (p ascending): the tail pieces of a short grid (GemmArgs::tail_main) summed into the
// slab region the missing blocks would have written
template <typename T>
__global__ void reduce_pieces_kernel(T *dst, int64_t ldd, const T *src, int64_t rows, int64_t cols, int npieces, int64_t stride, const int *done) {
    if (done && *done) return;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < rows * cols; e += (int64_t)gridDim.x * blockDim.x) {
        T acc = src[e];
        for (int q = 1; q < npieces; ++q) acc += src[e + q * stride];
        dst[(e % cols) + (e / cols) * ldd] = acc;
    }
}

// Split-K combine of a product that was launched a few blocks SHORT (GemmArgs::tail_main): the last slab's share of a small
// rectangle of the output arrived as `npieces` tail pieces instead.  ONE launch instead of reduce_pieces_kernel (pieces -> last slab)
// + reduce_slabs_kernel (slabs -> dst), and the same bits: dst = slab_0 + ... + slab_{n-2} + L, where L is the last slab's value
// or, inside the rectangle, piece_0 + piece_1 + ... (ascending; 8 loads in flight instead of a chain of dependent ones, which
// had made the stand-alone pieces pass the slowest small launch of a ProjectedALS iteration: 24 us for 0.5 MB).
//   mode 0: dst is flat (count elements), the rectangle is the contiguous range [off, off + plen): piece index e - off
//   mode 1: dst is ld x cols column-major (ld a multiple of 256), the rectangle is rows [r0, r0 + prow) of every column, pieces
//           are compact (ld = prow): piece index (i - r0) + a*prow
// V consecutive elements per thread as one access per slab / piece (V = 16 bytes' worth when every offset, length and stride is a
// multiple of it -- the launch sites check --, else 1)
template <typename T, int V = 1>
__global__ __launch_bounds__(256) void reduce_slabs_tail_kernel(T *dst, const T *src, int64_t count, int nslab, int64_t stride, const T *pieces,
                                                                int npieces, int64_t pstride, int mode, int64_t off, int64_t plen, int64_t ld,
                                                                int64_t r0, int64_t prow, const int *done) {
    NMFX_DONE_GUARD(done);
    typedef T vec_t __attribute__((ext_vector_type(V)));
    auto ldv = [](const T *p) { if constexpr (V == 1) return *p; else return *reinterpret_cast<const vec_t *>(p); };
    // Blocks walk the output BACKWARDS: the rectangle that arrives as pieces is the END of the output in both modes, and its few dozen blocks
    // carry as many bytes as all the others together -- dispatched last they ran alone behind everybody else (28 us for 48 MB of slabs at
    // 16384 x 256; dispatched first they run beside the light blocks)
    const unsigned bid = gridDim.x - 1u - blockIdx.x;
    int64_t e = ((int64_t)bid * 256 + threadIdx.x) * V;
    int64_t pidx = -1;
    if (mode == 0) {
        if (e >= count) return;
        if (e >= off && e < off + plen) pidx = e - off;
    } else {
        // column index fastest over the blocks: the blocks that carry the pieces (the last row blocks of EVERY column) are then
        // consecutive block ids, i.e. dealt to all 8 XCDs (row index fastest put them all on two XCDs: 48 us instead of 24)
        const unsigned cols = (unsigned)(count / ld), a = bid % cols, i = ((bid / cols) * 256u + threadIdx.x) * (unsigned)V;
        e = (int64_t)i + (int64_t)a * ld;
        if ((int64_t)i >= r0 && (int64_t)i < r0 + prow) pidx = ((int64_t)i - r0) + (int64_t)a * prow;
    }
    auto lastv = ldv(src);   // (type only; overwritten below)
    if (pidx >= 0) {
        auto acc = ldv(pieces + pidx);
        int q = 1;
        // the rectangle is a small part of the output, but every element of it is the sum of ~64 pieces: as many bytes as all the
        // slabs together, read by a few dozen blocks -- 16 loads in flight per thread (summed in piece order all the same)
        for (; q + 16 <= npieces; q += 16) {
            decltype(acc) v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = ldv(pieces + (int64_t)(q + u) * pstride + pidx);
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += v[u];
        }
        for (; q + 8 <= npieces; q += 8) {
            decltype(acc) v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ldv(pieces + (int64_t)(q + u) * pstride + pidx);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; q < npieces; ++q) acc += ldv(pieces + (int64_t)q * pstride + pidx);
        lastv = acc;
    } else {
        lastv = ldv(src + (int64_t)(nslab - 1) * stride + e);
    }
    auto out = lastv;
    if (nslab > 1) {
        auto s = ldv(src + e);
        for (int k = 1; k < nslab - 1; ++k) s += ldv(src + (int64_t)k * stride + e);
        out = s + lastv;
    }
    if constexpr (V == 1) dst[e] = out;
    else *reinterpret_cast<vec_t *>(dst + e) = out;
}

// ---- multdiv, single GPU: everything that follows a numerator product in ONE pass over the factor (src/multupd.jl:176-179, 188-191
// + stop_condition's sums, src/common.jl:95-104): split-K slabs summed (ascending, in T, like reduce_slabs_kernel), the scaling
// of div_update_kernel, the statistics of row_stats_kernel / col_stats_kernel and the OTHER side's next divisor (sum over the
// other axis of the NEW factor) -- same chunking and the same arithmetic as the separate kernels: bit-identical results, 4
// launches and 3 passes over the factor fewer per side.
//   partial[(chunk*K + j)*2 + {0,1}] = dev, sum;   partial[2*K*nchunks + chunk*K + j] = sum of the new factor's component j
template <typename T>
__global__ void div_h_fused_kernel(T *Hn, const T *Ho, const T *num, int nslab, int64_t slab_stride, const T *sW, int64_t k, int64_t n,
                                   int64_t cols, int64_t ld, int K, T lambda, double *partial, const int *done) {
    NMFX_DONE_GUARD(done);
    const int chunk = blockIdx.x, nchunks = gridDim.x;
    const int64_t per = (cols + nchunks - 1) / nchunks;
    const int64_t beg = chunk * per, end = (beg + per < cols) ? beg + per : cols;
    for (int j = threadIdx.x; j < K; j += blockDim.x) {
        double dev = 0.0, sum = 0.0, rs = 0.0;
        const T d = (j < k) ? sW[j] + lambda : (T)1;
        // 4 columns per trip: 4 x (nslab + 2) independent loads in flight per thread (one column per trip left the pass latency
        // bound: 61 us for 4 x 16 MB); the sums are still added in column order
        for (int64_t i0 = beg; i0 < end; i0 += 4) {
            T a[4], b[4], nu[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = i0 + u, o = j + i * ld;
                const bool in = i < end;
                b[u] = in ? Ho[o] : (T)0;
                a[u] = in ? Hn[o] : (T)0;                 // padding: whatever the buffer holds (zeros), never written
                nu[u] = (in && j < k && i < n) ? num[o] : (T)0;
            }
            for (int s = 1; s < nslab; ++s) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t i = i0 + u;
                    if (i < end && j < k && i < n) nu[u] += num[(int64_t)s * slab_stride + j + i * ld];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = i0 + u;
                if (i >= end) break;
                if (j < k && i < n) {
                    a[u] = b[u] * (nu[u] / d);
                    Hn[j + i * ld] = a[u];
                }
                const T df = a[u] - b[u], sp = a[u] + b[u];
                dev += (double)(T)(df * df);
                sum += (double)(T)(sp * sp);
                rs += (double)a[u];
            }
        }
        partial[((int64_t)chunk * K + j) * 2] = dev;
        partial[((int64_t)chunk * K + j) * 2 + 1] = sum;
        partial[(int64_t)2 * K * nchunks + (int64_t)chunk * K + j] = rs;
    }
}

// grid = (chunks, K): one block per (row chunk, component)
template <typename T>
__global__ void div_w_fused_kernel(T *Wn, const T *Wo, const T *num, int nslab, int64_t slab_stride, const T *sH, int64_t p, int64_t k,
                                   int64_t rows, int64_t ld, int K, T lambda, double *partial, const int *done) {
    NMFX_DONE_GUARD(done);
    __shared__ double sm[12];
    const int j = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const int64_t per = (rows + nchunks - 1) / nchunks;
    const int64_t beg = chunk * per, end = (beg + per < rows) ? beg + per : rows;
    const T d = (j < k) ? sH[j] + lambda : (T)1;
    double dev = 0.0, sum = 0.0, cs = 0.0;
    for (int64_t i = beg + threadIdx.x; i < end; i += blockDim.x) {
        const int64_t o = i + (int64_t)j * ld;
        const T b = Wo[o];
        T a = Wn[o];
        if (j < k && i < p) {
            T nu = num[o];
            for (int s = 1; s < nslab; ++s) nu += num[(int64_t)s * slab_stride + o];
            a = b * (nu / d);
            Wn[o] = a;
        }
        const T df = a - b, sp = a + b;
        dev += (double)(T)(df * df);
        sum += (double)(T)(sp * sp);
        cs += (double)a;
    }
    for (int off = 32; off > 0; off >>= 1) { dev += __shfl_down(dev, off, 64); sum += __shfl_down(sum, off, 64); cs += __shfl_down(cs, off, 64); }
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[w] = dev; sm[4 + w] = sum; sm[8 + w] = cs; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0, c = 0.0;
        for (int q = 0; q < nw; ++q) { a += sm[q]; b += sm[4 + q]; c += sm[8 + q]; }
        partial[((int64_t)chunk * K + j) * 2] = a;
        partial[((int64_t)chunk * K + j) * 2 + 1] = b;
        partial[(int64_t)2 * K * nchunks + (int64_t)chunk * K + j] = c;
    }
}

// statistics (2K doubles) and component sums (K values of T) of the fused kernels above in one launch; a wave per output, the
// chunk order of finalize_partials_kernel
template <typename T>
__global__ void finalize_div_kernel(const double *partial, int nchunks, int K, double *stat_out, T *sum_out, const int *done) {
    NMFX_DONE_GUARD(done);
    const int e = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (e >= 3 * K) return;
    const int lane = threadIdx.x & 63;
    const double *src = (e < 2 * K) ? partial + e : partial + (int64_t)2 * K * nchunks + (e - 2 * K);
    const int stride = (e < 2 * K) ? 2 * K : K;
    double s = 0.0;
    for (int c = lane; c < nchunks; c += 64) s += src[(int64_t)c * stride];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) {
        if (e < 2 * K) stat_out[e] = s;
        else sum_out[e - 2 * K] = (T)s;
    }
}

}  // namespace nmfx
