// cd.hpp -- row-parallel kernels of the two coordinate-descent updaters:
//   CoordinateDescent  _update_coord_descent!   src/coorddesc.jl:107-158
//   GreedyCD           _update_GreedyCD!        src/greedycd.jl:91-163
// Both updaters first form the k x k Gram P and the numerator Z with the big GEMMs (shared with the other algorithms) and
// then run a sequential-in-the-components sweep over every sample row -- and the rows are INDEPENDENT: row i only reads
// P, Z(i, :) and its own W(i, :).  One wavefront owns one sample row (CD: a few rows, to share the Gram-row loads); the
// k components live in registers, component r on lane r % 64 slot r / 64; dot products over the components are
// butterfly reductions (fixed order, so results do not depend on the launch).
//
// "Sample-major view": the updaters are written for W (p x k, element (i, t) at W[i + t*ld]) and are re-used for H through
// the transposed views Ht, X' (coorddesc.jl:168-172, greedycd.jl:170-174): element (j, t) of Ht is H[t + j*ld].  A view is
// (pointer, sample stride, component stride).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "kernels.hpp"

namespace nmfx {

template <typename T> struct SampleView {
    T *p;
    int64_t ss, cs;   // sample stride, component stride
    __device__ __forceinline__ T &at(int64_t i, int64_t t) const { return p[i * ss + t * cs]; }
};

template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
template <typename T> __device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T o = __shfl_xor(v, off, 64);
        v = (o > v) ? o : v;
    }
    return v;
}

// exactly-rounded single operations (no fused multiply-add contraction): the greedy sweep restates the reference's
// expressions operation by operation
// (HIP's __fmul_rn / __fadd_rn are plain `*` / `+`, which -ffp-contract=fast fuses across the call: the `contract(off)` pragma
// is what keeps  G(r) += S(q) P(q, r)  a rounded product followed by a rounded sum, as in the reference.)
__device__ __forceinline__ float op_mul(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ double op_mul(double a, double b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float op_add(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ double op_add(double a, double b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float op_sub(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ double op_sub(double a, double b) {
#pragma clang fp contract(off)
    return a - b;
}

// shuffle = true (src/coorddesc.jl:130-131): sweeping the components in the order perm[0], perm[1], ... is the in-order sweep of
// the problem with its components renamed -- W'(:, s) = W(:, perm[s]), Z'(:, s) = Z(:, perm[s]), P'(a, b) = P(perm[a], perm[b]) --
// so the sweep kernels below stay as they are and these two kernels rename on the way in and out.
template <typename T>
__global__ void permute_components_kernel(SampleView<T> dst, SampleView<const T> src, const int *__restrict__ perm, int64_t nsamples, int k,
                                          int inverse, const int *done) {
    NMFX_DONE_GUARD(done);
    const int64_t total = nsamples * k;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t i;
        int s;
        if (src.cs == 1) { s = (int)(e % k); i = e / k; }           // components contiguous (the H side's transposed view)
        else { i = e % nsamples; s = (int)(e / nsamples); }         // samples contiguous (the W side)
        if (inverse) dst.at(i, perm[s]) = src.at(i, s);
        else dst.at(i, s) = src.at(i, perm[s]);
    }
}
template <typename T>
__global__ void permute_gram_kernel(T *dst, const T *src, int64_t ld, const int *__restrict__ perm, int k, const int *done) {
    NMFX_DONE_GUARD(done);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < k * k; e += gridDim.x * blockDim.x) {
        const int a = e % k, b = e / k;
        dst[a + (int64_t)b * ld] = src[perm[a] + (int64_t)perm[b] * ld];
    }
}

// ---------------------------------------------------------------------------
// CoordinateDescent sweep (src/coorddesc.jl:130-156).  For every sample row i, components t = 1..k in order:
//     grad = -Z'(i,t) + sum_r P(t,r) W(i,r)          Z' = Z - l1 (:121-123), P already carries + l2 on its diagonal (:118-120)
//     hess = P(t,t);  if hess != 0:  W(i,t) = max(W(i,t) - grad/hess, 0)
// (The reference loops t outer / i inner; rows do not interact, so i outer / t inner gives the same W.  The scalar
// `violation` it accumulates is stored in the state but never read -- nmf_skeleton! stops on stop_condition -- and is not
// computed here.)  A wave owns R rows; the dot product is a 64-lane butterfly instead of the reference's left-to-right sum.
// ---------------------------------------------------------------------------
template <typename T, int KMAX, int R>
__global__ __launch_bounds__(256) void cd_sweep_kernel(SampleView<const T> Wold, SampleView<T> Wnew, SampleView<const T> Z,
                                                       const T *__restrict__ P, int64_t ldp, int64_t nsamples, int k, T l1,
                                                       const int *done) {
    NMFX_DONE_GUARD(done);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * R;
    if (i0 >= nsamples) return;
    const int km = (k + 63) / 64;
    T w[R][KMAX], z[R][KMAX];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < KMAX; ++m) {
            const int c = lane + 64 * m;
            const bool ok = (m < km) && (c < k) && (i0 + r < nsamples);
            w[r][m] = ok ? Wold.at(i0 + r, c) : (T)0;
            z[r][m] = ok ? (T)(Z.at(i0 + r, c) - l1) : (T)0;
        }
#pragma unroll
    for (int m0 = 0; m0 < KMAX; ++m0) {
        if (m0 < km) {
            const int tend = (k - 64 * m0 < 64) ? (k - 64 * m0) : 64;
            for (int tl = 0; tl < tend; ++tl) {
                const int t = 64 * m0 + tl;
                T a[KMAX];
#pragma unroll
                for (int m = 0; m < KMAX; ++m) a[m] = (m < km && lane + 64 * m < k) ? P[(int64_t)t * ldp + lane + 64 * m] : (T)0;
                const T hess = __shfl(a[m0], tl, 64);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    T part = (T)0;
#pragma unroll
                    for (int m = 0; m < KMAX; ++m) part += a[m] * w[r][m];
                    const T grad = wave_sum(part) - __shfl(z[r][m0], tl, 64);
                    const T wt = __shfl(w[r][m0], tl, 64);
                    T nw = wt - grad / hess;
                    nw = (nw > (T)0) ? nw : ((nw != nw) ? nw : (T)0);      // max(., zero(grad)); NaN propagates
                    if (hess != (T)0 && lane == tl) w[r][m0] = nw;
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < KMAX; ++m) {
            const int c = lane + 64 * m;
            if ((m < km) && (c < k) && (i0 + r < nsamples)) Wnew.at(i0 + r, c) = w[r][m];
        }
}

// value of the lane N places to the right inside the 16-lane DPP row (row_ror:N)
template <int N> __device__ __forceinline__ float row_ror(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, false));
}
template <int N> __device__ __forceinline__ double row_ror(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), 0x120 + N, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x120 + N, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Same sweep, 16 lanes per sample row (4 rows per wavefront): lane l of a row's group holds the KPL = K/16 consecutive
// components l*KPL .. l*KPL+KPL-1, so a Gram row is fetched with 16-byte loads, the dot product needs only a 4-step
// butterfly, and the four rows of a wave share every shuffle instruction (the 64-lane form above spends most of its
// issue slots in six-step reductions, one per row: 435 -> see DESIGN.md for the measured figure).  P must be K x K with
// zero padding (it is: Gram of zero-padded factors), so rows are read unguarded.
template <typename T, int KPLMAX>
__global__ __launch_bounds__(256) void cd_sweep16_kernel(SampleView<const T> Wold, SampleView<T> Wnew, SampleView<const T> Z,
                                                         const T *__restrict__ P, int64_t ldp, int64_t nsamples, int k, int kpl, T l1,
                                                         const int *done) {
    NMFX_DONE_GUARD(done);
    constexpr int VEC = 16 / sizeof(T);
    using vec_t = T __attribute__((ext_vector_type(VEC)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, l = lane & 15;
    const int64_t i = ((int64_t)blockIdx.x * 4 + wave) * 4 + g;
    const bool live = i < nsamples;
    T w[KPLMAX], z[KPLMAX];
#pragma unroll
    for (int s = 0; s < KPLMAX; ++s) {
        const int c = l * kpl + s;
        const bool ok = live && (s < kpl) && (c < k);
        w[s] = ok ? Wold.at(i, c) : (T)0;
        z[s] = ok ? (T)(Z.at(i, c) - l1) : (T)0;
    }
    for (int lt = 0; lt < 16; ++lt) {
        if (lt * kpl >= k) break;
#pragma unroll
        for (int s = 0; s < KPLMAX; ++s) {
            const int t = lt * kpl + s;
            if (s < kpl && t < k) {
                T a[KPLMAX];
                const T *prow = P + (int64_t)t * ldp + l * kpl;
#pragma unroll
                for (int v = 0; v < KPLMAX; v += VEC) {
                    if (v < kpl) {
                        const vec_t pv = *reinterpret_cast<const vec_t *>(prow + v);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) a[v + e] = pv[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) a[v + e] = (T)0;
                    }
                }
                T part = (T)0;
#pragma unroll
                for (int v = 0; v < KPLMAX; ++v) part += a[v] * w[v];
                // 16-lane all-reduce by DPP row rotations (a DPP row IS a sample row's 16 lanes): the same pairs as the xor butterfly
                // (rotating by 8, 4, 2, 1 adds partials that are already periodic in 8, 4, 2), i.e. the same bits, without four
                // ds_bpermute round trips through the LDS crossbar on the dependency chain of every coordinate
                part += row_ror<8>(part);
                part += row_ror<4>(part);
                part += row_ror<2>(part);
                part += row_ror<1>(part);
                // only the lane that owns coordinate t keeps the result, and it holds everything the update needs itself:
                // a[s] = P(t, t), z[s] = Z(i, t) - l1, w[s] = W(i, t).  The other lanes compute garbage that is never stored
                // (three more broadcasts per coordinate saved).
                const T hess = a[s];
                const T grad = part - z[s];
                T nw = w[s] - grad / hess;
                nw = (nw > (T)0) ? nw : ((nw != nw) ? nw : (T)0);
                if (hess != (T)0 && l == lt) w[s] = nw;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < KPLMAX; ++s) {
        const int c = l * kpl + s;
        if (live && (s < kpl) && (c < k)) Wnew.at(i, c) = w[s];
    }
}

// g / den, correctly rounded, with den loop-invariant.  Float32: (float)((double)g * (1.0 / den)) IS the correctly rounded quotient --
// the Float64 product is within 2^-52 of g / den, while a quotient of two 24-bit significands that is not itself a float stays at
// least 2^-49 (relative) away from every rounding boundary of the float grid (g - m den is a non-zero multiple of the last place
// of the 49-bit product m den), so the final rounding cannot go the other way; infinities, NaN and signed zeros behave like the
// division (tests/test_host_api.py::test_greedy_division_identity checks 4e6 operand pairs incl. adversarial ones).  3 operations on
// the greedy step's dependency chain instead of the ~12 of v_div_scale / v_rcp / fma x5 / v_div_fmas / v_div_fixup.  Float64 divides.
__device__ __forceinline__ float greedy_div(float g, float, double rden) { return (float)((double)g * rden); }
// The same quotient from Float32 operations only (round 6; the register form of the sweep): with r = RN(1 / den) -- ONE IEEE division per
// component and row --  q0 = RN(g r),  e = g - den q0 (one fused multiply-add: the residual of a quotient that is within an ulp is
// exactly representable or rounds harmlessly),  q = RN(q0 + e r)  is the correctly rounded g / den (Markstein's theorem for a correctly
// rounded reciprocal; tests/test_host_api.py::test_greedy_division_fma_form checks it in exact rational arithmetic on random and on
// adversarial operands -- all-ones significands, quotients next to rounding boundaries).  v_mul_f32 + 2 v_fma_f32 at 1.1 ns each
// where v_cvt_f64_f32 + v_mul_f64 + v_cvt_f32_f64 take 1.8 + 1.9 + 1.8 ns with eight waves on a SIMD (profiles/r04_valu_rate_probe.log):
// -9 ns on a greedy step of ~160.  Non-finite g (the lanes beyond k carry +inf) gives NaN here where the Float64 form gives inf: either way
// the lane's D is NaN, which the arg-max skips.
__device__ __forceinline__ float greedy_div_fma(float g, float den, float r) {
    float q0, e, q;
    asm("v_mul_f32 %0, %1, %2" : "=v"(q0) : "v"(g), "v"(r));
    asm("v_fma_f32 %0, -%1, %2, %3" : "=v"(e) : "v"(den), "v"(q0), "v"(g));
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(q) : "v"(e), "v"(r), "v"(q0));
    return q;
}
__device__ __forceinline__ double greedy_div(double g, double den, double) { return g / den; }

// ---------------------------------------------------------------------------
// Blocked sweep (Float32, big problems).  The sweeps above are one long dependency chain per sample row: every coordinate
// recomputes its gradient as a K-long dot product with the current row (a Gram row fetched from L2, a 16- or 64-lane reduction,
// a division), 256 times in a row at k = 256 -- 250 us per side at 16384 rows, 10x what the arithmetic costs.  Here the
// coordinates are taken 16 at a time.  For a block B of 16 coordinates the part of the gradient that does not change while B is
// being swept,
//     G_B = W_tile * P[:, B] - Z_B          (64 sample rows x K) * (K x 16)
// is ONE matrix-core product per block (v_mfma_f32_16x16x4, 64 instructions per wave), and the sweep inside the block only needs
// the 16 x 16 diagonal block of P:  w_t <- max(0, w_t - g_t / P_tt),  g_t' += P_t't * (w_t_new - w_t_old) for the other t' of the
// block -- the same Gauss-Seidel sweep in exact arithmetic (each g_t is the full gradient at the moment coordinate t is updated),
// a different summation order in floating point.  A workgroup owns 64 sample rows; the tile of W lives in LDS for the whole
// sweep (updated block by block), four lanes share a sample row in the in-block sweep (4 coordinates each, the step broadcast
// inside the quad by DPP).  Operand fragments: lane (i, kg) holds 4 consecutive contraction indices, MFMA q of a group of 16
// contracts indices {q, 4 + q, 8 + q, 12 + q} of the group (any partition is a valid order of the sum), so both operands are
// 16-byte reads -- W from LDS, P straight from global memory (P is symmetric: column c of the product is row c of P).
template <int SEL> __device__ __forceinline__ float quad_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), SEL * 0x55, 0xf, 0xf, false));
}
template <typename T> struct Mfma16;
template <> struct Mfma16<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
};
template <> struct Mfma16<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
};
template <int N, typename F, int... I> __device__ __forceinline__ void cd_static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void cd_static_for(F &&f) { cd_static_for_impl<N>(static_cast<F &&>(f), std::make_integer_sequence<int, N>{}); }

constexpr int CD_BLK_GS = 20;
// ROWS sample rows per workgroup (64: Float32; Float64: 64 up to k = 256, 32 beyond -- the tile of W must fit LDS), 4 lanes per row
template <typename T> static inline size_t cd_blocked_lds_bytes(int64_t K, int rows) {
    return (size_t)(rows * (K + 16 / sizeof(T)) + rows * CD_BLK_GS + 16 * CD_BLK_GS) * sizeof(T) + 16 * sizeof(double);
}
template <int SEL> __device__ __forceinline__ double quad_bcast(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), SEL * 0x55, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), SEL * 0x55, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float cd_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double cd_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// KG = K / (4 VEC) groups of contraction indices, VEC = 16 bytes' worth of T (compile time: the P fragments of a whole block, KG
// 16-byte loads per lane, are requested BEFORE the in-block sweep of the previous block and consumed after it -- with one wave per
// SIMD nothing else hides their L2 round trips).  Lane (i, kg) holds VEC consecutive indices of a group; MFMA q contracts indices
// {q, VEC + q, 2 VEC + q, 3 VEC + q} of the group.
template <typename T, int KG, int ROWS>
__global__ __launch_bounds__(ROWS * 4) void cd_sweep_blocked_kernel(SampleView<const T> Wold, SampleView<T> Wnew, SampleView<const T> Z,
                                                                    const T *__restrict__ P, int64_t ldp, int64_t nsamples, int k, T l1,
                                                                    const int *done) {
    NMFX_DONE_GUARD(done);
    constexpr int VEC = 16 / sizeof(T), GR = 4 * VEC, K = GR * KG, LDW = K + VEC, NT = ROWS * 4;
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    typedef T acc_t __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char cd_lds_raw[];
    T *Ws = reinterpret_cast<T *>(cd_lds_raw);     // [ROWS][LDW]      the tile of W, component-contiguous
    T *Gs = Ws + ROWS * LDW;                       // [ROWS][20]       W_tile * P[:, B]
    T *Pt = Gs + ROWS * CD_BLK_GS;                 // [16 t][20]       P(c0 + t, c0 + t')
    double *Pr = reinterpret_cast<double *>(Pt + 16 * CD_BLK_GS);   // [16 t]   1 / P(c0 + t, c0 + t) in Float64 (greedy_div, Float32 only)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kg = lane >> 4;
    const int64_t i0 = (int64_t)blockIdx.x * ROWS;
    const bool rows_contig = (Wold.ss == 1);       // W side: consecutive sample rows are consecutive in memory; H side: components are
    // (every load of the tile in flight at once: batches of 8 cost one HBM round trip each, ~15 us in all)
    {
        constexpr int NL = ROWS * K / NT;
        T v[NL];
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            const int idx = tid + NT * q;
            const int r = rows_contig ? (idx % ROWS) : (idx / K), c = rows_contig ? (idx / ROWS) : (idx - (idx / K) * K);
            v[q] = (i0 + r < nsamples && c < k) ? Wold.at(i0 + r, c) : (T)0;
        }
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            const int idx = tid + NT * q;
            const int r = rows_contig ? (idx % ROWS) : (idx / K), c = rows_contig ? (idx / ROWS) : (idx - (idx / K) * K);
            Ws[r * LDW + c] = v[q];
        }
    }
    __syncthreads();
    const int rr = tid >> 2, part = tid & 3;       // in-block sweep: sample row rr, coordinates 4 part .. 4 part + 3 of the block
    const bool live = i0 + rr < nsamples;
    vec_t bf[KG];
    auto load_p = [&](int c0) {
        const T *prow = P + (int64_t)(c0 + i) * ldp + VEC * kg;
#pragma unroll
        for (int j = 0; j < KG; ++j) bf[j] = *reinterpret_cast<const vec_t *>(prow + GR * j);
    };
    load_p(0);
    // this row's Z and the diagonal block of P are requested ONE BLOCK AHEAD (Z comes from HBM: its round trip is longer than a product)
    constexpr int ND = 256 / NT;                   // elements of the 16 x 16 diagonal block per thread (1, or 2 with 32-row tiles)
    T z4n[4];
    T pdn[ND];
    auto load_zd = [&](int c0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) z4n[u] = (live && c0 + 4 * part + u < k) ? Z.at(i0 + rr, c0 + 4 * part + u) : (T)0;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const int e = tid + NT * d;
            pdn[d] = P[(int64_t)(c0 + (e >> 4)) * ldp + c0 + (e & 15)];
        }
    };
    load_zd(0);
    for (int c0 = 0; c0 < k; c0 += 16) {
        T z4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) z4[u] = (live && c0 + 4 * part + u < k) ? (T)(z4n[u] - l1) : (T)0;
        T pdiag[ND];
#pragma unroll
        for (int d = 0; d < ND; ++d) pdiag[d] = pdn[d];
        if (c0 + 16 < k) load_zd(c0 + 16);
        // G_B tile of this wave: rows 16 wave .. + 15
        acc_t acc = {(T)0, (T)0, (T)0, (T)0}, acc2 = {(T)0, (T)0, (T)0, (T)0};
        const T *arow = Ws + (16 * wave + i) * LDW + VEC * kg;
#pragma unroll
        for (int j = 0; j < KG; ++j) {
            const vec_t a = *reinterpret_cast<const vec_t *>(arow + GR * j), b = bf[j];
#pragma unroll
            for (int q = 0; q < VEC; q += 2) {
                acc = Mfma16<T>::mma(a[q], b[q], acc);
                acc2 = Mfma16<T>::mma(a[q + 1], b[q + 1], acc2);
            }
        }
        if (c0 + 16 < k) load_p(c0 + 16);      // lands during the in-block sweep below
        acc += acc2;
        // D layout: Float32 lane (n = i, mg = kg) holds rows 4 mg + r of column n; Float64 rows mg + 4 r
#pragma unroll
        for (int r = 0; r < 4; ++r) Gs[(16 * wave + ((sizeof(T) == 4) ? 4 * kg + r : kg + 4 * r)) * CD_BLK_GS + i] = acc[r];
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const int e = tid + NT * d;
            Pt[(e >> 4) * CD_BLK_GS + (e & 15)] = pdiag[d];
            if (sizeof(T) == 4 && (e >> 4) == (e & 15)) Pr[e & 15] = 1.0 / (double)pdiag[d];
        }
        __syncthreads();
        T g4[4], w4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            g4[u] = Gs[rr * CD_BLK_GS + 4 * part + u] - z4[u];
            w4[u] = Ws[rr * LDW + c0 + 4 * part + u];
        }
        cd_static_for<16>([&](auto TC) {
            constexpr int t = decltype(TC)::value, owner = t >> 2, u = t & 3;
            const T hess = Pt[t * CD_BLK_GS + t];
            const T wt = w4[u];
            T nw = wt - greedy_div(g4[u], hess, Pr[t]);                                        // = g / hess, correctly rounded (only the owner's value is used)
            nw = (nw > (T)0) ? nw : ((nw != nw) ? nw : (T)0);                                  // max(., zero(grad)); NaN propagates
            nw = (hess != (T)0) ? nw : wt;
            const T delta = quad_bcast<owner>((T)(nw - wt));
            if (part == owner) w4[u] = nw;
            // four scalar FMAs, not a vector `g4 += pv * delta`: that compiles to v_pk_mul_f32 with delta in a register PAIR whose
            // other half the allocator gave to a prefetch still in flight -- s_waitcnt vmcnt(0) in front of the first step, i.e. the
            // next block's P fragments and Z waited for at once (+2 us per block)
#pragma unroll
            for (int e = 0; e < 4; ++e) g4[e] = cd_fma(Pt[t * CD_BLK_GS + 4 * part + e], delta, g4[e]);   // P(c0 + t, c0 + 4 part + e)
        });
#pragma unroll
        for (int u = 0; u < 4; ++u) Ws[rr * LDW + c0 + 4 * part + u] = w4[u];
        __syncthreads();
    }
    for (int idx = tid; idx < ROWS * K; idx += NT) {
        const int r = rows_contig ? (idx % ROWS) : (idx / K), c = rows_contig ? (idx / ROWS) : (idx - (idx / K) * K);
        if (i0 + r < nsamples && c < k) Wnew.at(i0 + r, c) = Ws[r * LDW + c];
    }
}

// ---------------------------------------------------------------------------
// GreedyCD (src/greedycd.jl:91-163).  Per sample row i (registers: W, G, S, D rows):
//     S(r) = max(0, W(r) - G(r)/(eps + P(r,r))) - W(r);   D(r) = -G(r) S(r) - 0.5 P(r,r) S(r)^2        (:120-125, :150-153)
//     q = argmax_r D(r)  (first index on ties, like Julia's argmax)
// p_init = max over ALL rows of D(i, q_i), floor -1 (:127-132)   -> greedy_pinit_kernel + one tiny reduction
// then at most k^2 steps per row (:137-158): stop when D(q) < nu * p_init;  Wnew(q) += S(q);  G(r) += S(q) P(q, r);
// recompute S, D;  q = argmax.   Finally W = max(W + Wnew, 0) (:160-161).
// G arrives as W*P - Z from the GEMM; + lambda (:113-115) is applied on load.
// ---------------------------------------------------------------------------
// max(zero(T), t) of the greedy step.  Float32: ONE integer maximum on the bits -- negative values and -0 have the sign bit set (a
// negative integer) and become +0, positive values and NaN are positive integers and pass (Julia's max(0, NaN) is NaN too; a NaN
// with the sign bit set would become 0, which changes nothing: its component's G is NaN, so its D is, and NaN components are never
// picked) -- instead of a comparison, the wait states of its mask, and a select.
__device__ __forceinline__ float clamp0(float t) { const int b = __float_as_int(t); return __int_as_float(b < 0 ? 0 : b); }
__device__ __forceinline__ double clamp0(double t) { return (t <= 0.0) ? 0.0 : t; }

template <typename T> __device__ __forceinline__ void greedy_sd(T w, T g, T prr, T den, double rden, T &s, T &d) {
    const T t = clamp0(op_sub(w, greedy_div(g, den, rden)));   // max(zero(T), .)
    s = op_sub(t, w);
    d = op_sub(op_mul(-g, s), op_mul(op_mul((T)0.5, prr), op_mul(s, s)));
}

// Float32 single operations as asm statements: exactly one rounded operation each (nothing to contract), and out of reach of the SLP
// vectoriser, which pairs the slots of the greedy step into v_pk_mul/add_f32 and then spends a v_mov per operand re-pairing them
// (10 moves per step in the k = 256 form; the plain v_mul/v_add/v_sub issue at 1.1 ns, a v_pk at 1.8: nothing to gain from pairs).
__device__ __forceinline__ float f32_mul(float a, float b) { float r; asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float f32_mul_s(float a_uniform, float b) { float r; asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "s"(a_uniform), "v"(b)); return r; }
__device__ __forceinline__ float f32_negmul(float a, float b) { float r; asm("v_mul_f32_e64 %0, -%1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }   // (-a) * b
__device__ __forceinline__ float f32_add(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float f32_sub(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// S(q) -> a wave-uniform value, and Wnew(q) += S(q) in the owner lane, by a wave-uniform switch over the owner SLOT q / 64: one
// v_readlane and one v_add under the owner lane's mask (the generic form selects S's slot with KMAX - 1 vector selects and updates
// Wnew with KMAX adds and KMAX selects: 17 vector instructions of the ~85 of a k = 256 step).  The two scalar instructions between
// the v_readlane and the v_add are the wait states a VALU read of a VALU-written SGPR needs.
// (ONE asm statement with its own branches: written as a C++ switch, the merge of the cases comes back as vector moves or selects per
// slot -- the instructions this form is there to remove.  All paths meet at the last label; exec is restored there.)
#define NMFX_TAKE_CASE(S, W) "v_readlane_b32 %[sq], " S ", %[ql]\n\ts_nop 1\n\tv_add_f32 " W ", %[sq], " W "\n\t"
__device__ __forceinline__ float greedy_take(const float (&s)[1], float (&w)[1], int, int ql) {
    float sq; unsigned long long sv;
    asm volatile("s_mov_b64 %[sv], exec\n\ts_lshl_b64 exec, 1, %[ql]\n\t" NMFX_TAKE_CASE("%[s0]", "%[w0]") "s_mov_b64 exec, %[sv]"
                 : [sq] "=&s"(sq), [sv] "=&s"(sv), [w0] "+v"(w[0]) : [s0] "v"(s[0]), [ql] "s"(ql) : "scc");
    return sq;
}
__device__ __forceinline__ float greedy_take(const float (&s)[2], float (&w)[2], int qm, int ql) {
    float sq; unsigned long long sv;
    asm volatile("s_mov_b64 %[sv], exec\n\ts_lshl_b64 exec, 1, %[ql]\n\t"
                 "s_cmp_eq_u32 %[qm], 0\n\ts_cbranch_scc0 1f\n\t"
                 NMFX_TAKE_CASE("%[s0]", "%[w0]") "s_branch 4f\n"
                 "1:\n\t" NMFX_TAKE_CASE("%[s1]", "%[w1]")
                 "4:\n\ts_mov_b64 exec, %[sv]"
                 : [sq] "=&s"(sq), [sv] "=&s"(sv), [w0] "+v"(w[0]), [w1] "+v"(w[1]) : [s0] "v"(s[0]), [s1] "v"(s[1]), [qm] "s"(qm), [ql] "s"(ql) : "scc");
    return sq;
}
__device__ __forceinline__ float greedy_take(const float (&s)[3], float (&w)[3], int qm, int ql) {
    float sq; unsigned long long sv;
    asm volatile("s_mov_b64 %[sv], exec\n\ts_lshl_b64 exec, 1, %[ql]\n\t"
                 "s_cmp_lt_u32 %[qm], 2\n\ts_cbranch_scc0 2f\n\t"
                 "s_cmp_eq_u32 %[qm], 0\n\ts_cbranch_scc0 1f\n\t"
                 NMFX_TAKE_CASE("%[s0]", "%[w0]") "s_branch 4f\n"
                 "1:\n\t" NMFX_TAKE_CASE("%[s1]", "%[w1]") "s_branch 4f\n"
                 "2:\n\t" NMFX_TAKE_CASE("%[s2]", "%[w2]")
                 "4:\n\ts_mov_b64 exec, %[sv]"
                 : [sq] "=&s"(sq), [sv] "=&s"(sv), [w0] "+v"(w[0]), [w1] "+v"(w[1]), [w2] "+v"(w[2])
                 : [s0] "v"(s[0]), [s1] "v"(s[1]), [s2] "v"(s[2]), [qm] "s"(qm), [ql] "s"(ql) : "scc");
    return sq;
}
__device__ __forceinline__ float greedy_take(const float (&s)[4], float (&w)[4], int qm, int ql) {
    float sq; unsigned long long sv;
    asm volatile("s_mov_b64 %[sv], exec\n\ts_lshl_b64 exec, 1, %[ql]\n\t"
                 "s_cmp_lt_u32 %[qm], 2\n\ts_cbranch_scc0 2f\n\t"
                 "s_cmp_eq_u32 %[qm], 0\n\ts_cbranch_scc0 1f\n\t"
                 NMFX_TAKE_CASE("%[s0]", "%[w0]") "s_branch 4f\n"
                 "1:\n\t" NMFX_TAKE_CASE("%[s1]", "%[w1]") "s_branch 4f\n"
                 "2:\n\ts_cmp_eq_u32 %[qm], 2\n\ts_cbranch_scc0 3f\n\t"
                 NMFX_TAKE_CASE("%[s2]", "%[w2]") "s_branch 4f\n"
                 "3:\n\t" NMFX_TAKE_CASE("%[s3]", "%[w3]")
                 "4:\n\ts_mov_b64 exec, %[sv]"
                 : [sq] "=&s"(sq), [sv] "=&s"(sv), [w0] "+v"(w[0]), [w1] "+v"(w[1]), [w2] "+v"(w[2]), [w3] "+v"(w[3])
                 : [s0] "v"(s[0]), [s1] "v"(s[1]), [s2] "v"(s[2]), [s3] "v"(s[3]), [qm] "s"(qm), [ql] "s"(ql) : "scc");
    return sq;
}
#undef NMFX_TAKE_CASE

// Wave-wide maximum by DPP row shifts + row broadcasts (an inclusive scan: the total lands in lane 63) instead of ds_bpermute round
// trips through the LDS crossbar -- this reduction sits on the dependency chain of EVERY greedy step.  The result is wave-uniform
// (SGPRs): the step's control flow, the row of P it loads and the lane that owns S(q) are scalar from here on.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ int dpp_mov(int ident, int v) {
    return __builtin_amdgcn_update_dpp(ident, v, CTRL, ROW_MASK, 0xf, false);
}
// v of lane `l` (wave-uniform l): v_readlane with a scalar lane select
__device__ __forceinline__ float lane_read(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ double lane_read(double v, int l) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float lane63(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
__device__ __forceinline__ double lane63(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// (v never NaN: callers only feed values that compared greater than something)
template <int CTRL, int ROW_MASK> __device__ __forceinline__ void max_dpp_step(float &v) {
    const float ov = __int_as_float(dpp_mov<CTRL, ROW_MASK>(__float_as_int(-INFINITY), __float_as_int(v)));
    v = (ov > v) ? ov : v;
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ void max_dpp_step(double &v) {
    const long long ident = __double_as_longlong(-INFINITY), b = __double_as_longlong(v);
    const int lo = dpp_mov<CTRL, ROW_MASK>((int)(ident & 0xffffffffll), (int)(b & 0xffffffffll));
    const int hi = dpp_mov<CTRL, ROW_MASK>((int)(ident >> 32), (int)(b >> 32));
    const double ov = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    v = (ov > v) ? ov : v;
}
// Float32: the maximum and the DPP lane shift in ONE instruction (v_max_f32_dpp; lanes without a source lane are disabled by the DPP
// bound check and keep their value, which is what the -inf identity of the generic form achieves with a move, a compare and a
// select).  The compiler does not form it from the generic code; the s_nop covers the VALU-write -> DPP-read hazard, which its
// hazard recogniser cannot see inside an asm statement.  Same value as the generic form for non-NaN inputs (signed zeros compare
// equal everywhere the result is used).
__device__ __forceinline__ float wave_max_uniform(float v) {
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(v));
    return lane63(v);
}
template <typename T> __device__ __forceinline__ T wave_max_uniform(T v) {
    max_dpp_step<0x111, 0xf>(v);
    max_dpp_step<0x112, 0xf>(v);
    max_dpp_step<0x114, 0xf>(v);
    max_dpp_step<0x118, 0xf>(v);
    max_dpp_step<0x142, 0xa>(v);
    max_dpp_step<0x143, 0xc>(v);
    return lane63(v);
}

// max(a, b) / max(a, b, c) that SKIP NaN operands (v_max_f32 / v_max_f64 in IEEE mode return the non-NaN operand)
__device__ __forceinline__ float max_skip_nan(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float max3_skip_nan(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ double max_skip_nan(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double max3_skip_nan(double a, double b, double c) { return max_skip_nan(max_skip_nan(a, b), c); }

template <typename T, int KMAX> struct GreedyRow {
    T w[KMAX], g[KMAX], s[KMAX], d[KMAX], prr[KMAX], den[KMAX];   // den = eps + P(r, r) (greedycd.jl:121, :151)
    double rden[KMAX];                                              // 1 / den (Float32 rows only; dead code for Float64)
    float r32[KMAX];                                                // RN(1 / den) in Float32 (greedy_div_fma; Float32 rows only)
    __device__ __forceinline__ void load(const SampleView<const T> &W, const SampleView<const T> &G, const T *P, int64_t ldp,
                                         int64_t i, int k, int km, int lane, T lambda, T epsT) {
        (void)km;
#pragma unroll
        for (int m = 0; m < KMAX; ++m) {
            // Round 6: lane l owns the KMAX CONSECUTIVE components KMAX l .. KMAX l + KMAX - 1 (slot m <-> component KMAX l + m), not
            // l, l + 64, ...: the row P(q, :) a step needs is then ONE 16-byte load per lane (KMAX = 4, Float32) instead of four 4-byte
            // ones -- the address path of the CU (one instruction of 64 lanes at a time, whatever its width) is what the sweep's eight
            // waves per SIMD queue for.  Which lane holds which component changes no arithmetic.
            const int c = KMAX * lane + m;
            const bool ok = c < k;
            w[m] = ok ? W.at(i, c) : (T)0;
            // a lane without a component (c >= k) carries G = +inf: S = max(0, 0 - inf) - 0 = 0 and D = -inf * 0 - ... = NaN, now and after
            // every step (its entries of P's rows are the zero padding: inf + S(q) * 0 = inf), and NaN is what the arg-max skips -- the
            // FULL forms need no validity masks for it
            g[m] = ok ? G.at(i, c) : (T)INFINITY;
            if (ok && lambda > (T)0) g[m] = op_add(g[m], lambda);
            prr[m] = ok ? P[(int64_t)c * ldp + c] : (T)1;
            den[m] = op_add(epsT, prr[m]);
            rden[m] = 1.0 / (double)den[m];
            r32[m] = 1.0f / (float)den[m];
            greedy_sd(w[m], g[m], prr[m], den[m], rden[m], s[m], d[m]);
        }
    }
    // arg-max of D with the first index on ties: the VALUE by a DPP max reduction (2 operations per step instead of the 7 of a
    // (value, index) reduction -- this sits on every greedy step's dependency chain), then the index from wave ballots: a slot's lowest
    // set lane l gives its smallest component index c = KMAX l + m, and the smallest of the slots' candidates is the first index
    // FULL: ceil(k / 64) == KMAX -- the lanes beyond k hold D = NaN (see load): no validity masks
    template <bool FULL = false> __device__ __forceinline__ void argmax(int k, int km, int lane, T &best, int &q) const {
        (void)km;
        best = -INFINITY;
        if constexpr (FULL) {
            // `d > best ? d : best` from -inf == the NaN-skipping maximum (v_max: a NaN operand yields the other one), and v_max3 takes
            // two slots per instruction; signed zeros compare equal everywhere the value is used
#pragma unroll
            for (int m = 0; m + 1 < KMAX; m += 2) best = max3_skip_nan(best, d[m], d[m + 1]);
            if constexpr (KMAX & 1) best = max_skip_nan(best, d[KMAX - 1]);
        } else {
#pragma unroll
            for (int m = 0; m < KMAX; ++m) {
                const int c = KMAX * lane + m;
                const bool take = (c < k) && (d[m] > best);
                best = take ? d[m] : best;
            }
        }
        best = wave_max_uniform(best);
        // (s_ff1 of an empty ballot is -1: KMAX * -1 + m as an unsigned number is larger than every index, so the unsigned minimum skips it)
        unsigned qu = 0x7fffffffu;
#pragma unroll
        for (int m = 0; m < KMAX; ++m) {
            const int c = KMAX * lane + m;
            const unsigned long long hit = FULL ? __builtin_amdgcn_ballot_w64(d[m] == best) : __builtin_amdgcn_ballot_w64((c < k) && (d[m] == best));
            const unsigned cand = (unsigned)(KMAX * (__ffsll((long long)hit) - 1) + m);
            qu = (cand < qu) ? cand : qu;
        }
        q = (int)qu;
    }
};

// per-block maxima of D(i, q_i)  ->  part[blockIdx.x]
template <typename T, int KMAX>
__global__ __launch_bounds__(256) void greedy_pinit_kernel(SampleView<const T> W, SampleView<const T> G, const T *__restrict__ P,
                                                           int64_t ldp, int64_t nsamples, int k, T lambda, T epsT, T *part,
                                                           const int *done) {
    NMFX_DONE_GUARD(done);
    __shared__ T sm[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int km = (k + 63) / 64;
    T best = (T)-1;   // p_init = convert(T, -1.0)
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    if (i < nsamples) {
        GreedyRow<T, KMAX> row;
        row.load(W, G, P, ldp, i, k, km, lane, lambda, epsT);
        T v; int q;
        row.argmax(k, km, lane, v, q);
        best = (v > best) ? v : best;
    }
    if (lane == 0) sm[wave] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        T b = sm[0];
        for (int wv = 1; wv < 4; ++wv) b = (sm[wv] > b) ? sm[wv] : b;
        part[blockIdx.x] = b;
    }
}

// pinit[0] = max(part[0..n))  (one block)
template <typename T> __global__ void greedy_pinit_reduce_kernel(const T *part, int n, T *pinit, int *queue, const int *done) {
    NMFX_DONE_GUARD(done);
    __shared__ T sm[4];
    if (queue != nullptr && threadIdx.x < 32) queue[threadIdx.x * 32] = 0;   // the sweep's row counters (GREEDY_NQ x GREEDY_QSTRIDE, declared below)
    T b = (T)-1;
    for (int i = threadIdx.x; i < n; i += blockDim.x) b = (part[i] > b) ? part[i] : b;
    b = wave_max(b);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int wv = 1; wv < (int)(blockDim.x >> 6); ++wv) b = (sm[wv] > b) ? sm[wv] : b;
        pinit[0] = b;
    }
}

// The sweep of one sample row by one wave; `fetch(q, m)` returns P(q, lane + 64 m) (q wave-uniform).
template <typename T, int KMAX, bool FULL, typename Fetch>
__device__ __forceinline__ long long greedy_sweep_row(SampleView<const T> Wold, SampleView<T> Wout, SampleView<const T> G, const T *__restrict__ P, int64_t ldp,
                                                      int64_t i, int k, T lambda, T epsT, const T *pinit, int lane, Fetch fetch) {
    const int km = (k + 63) / 64;
    GreedyRow<T, KMAX> row;
    row.load(Wold, G, P, ldp, i, k, km, lane, lambda, epsT);
    T wnew[KMAX];
#pragma unroll
    for (int m = 0; m < KMAX; ++m) wnew[m] = (T)0;
    T hprr[KMAX];                                   // 0.5 * P(r, r): the first product of  0.5 * P(r, r) * S(r)^2  (:124, :153), loop-invariant
#pragma unroll
    for (int m = 0; m < KMAX; ++m) hprr[m] = op_mul((T)0.5, row.prr[m]);
    const T thresh = op_mul((T)0.001, pinit[0]);   // nu * p_init
    T dq; int q;
    row.template argmax<FULL>(k, km, lane, dq, q);
    const int max_steps = k * k;                    // k <= 1024 in this register form
    int step = 0;
    for (; step < max_steps; ++step) {
        if (dq < thresh) break;                       // wave-uniform (dq, q come out of wave_argmax as scalars)
        // S(q): owned by lane q % 64, slot q / 64
        const int ql = q / KMAX, qm = q % KMAX;       // owner lane and slot of component q (KMAX is a compile-time constant)
        constexpr bool TAKE = FULL && KMAX <= 4 && std::is_same<T, float>::value;
        T sq;
        T pq[KMAX];
        if constexpr (FULL) {
            // all KMAX loads of the row P(q, :) in flight at once, and issued first: their latency runs under the owner-slot switch
            // (the per-slot `m < km` test of the general form below is a branch per slot, each with its own load -> wait -> compute
            // round trip)
#pragma unroll
            for (int m = 0; m < KMAX; ++m) pq[m] = fetch(q, m);
        }
        if constexpr (TAKE) {
            sq = greedy_take(row.s, wnew, qm, ql);
        } else {
            T sq_owner = row.s[0];
#pragma unroll
            for (int m = 1; m < KMAX; ++m) sq_owner = (m == qm) ? row.s[m] : sq_owner;
            sq = lane_read(sq_owner, ql);
        }
        if constexpr (FULL) {
            if constexpr (!TAKE) {
#pragma unroll
                for (int m = 0; m < KMAX; ++m) wnew[m] = (m == qm && lane == ql) ? op_add(wnew[m], sq) : wnew[m];
            }
            T t[KMAX];
            if constexpr (std::is_same<T, float>::value) {
#pragma unroll
                for (int m = 0; m < KMAX; ++m) {
                    row.g[m] = f32_add(row.g[m], f32_mul_s(sq, pq[m]));
                    t[m] = f32_sub(row.w[m], greedy_div_fma(row.g[m], row.den[m], row.r32[m]));
                }
#pragma unroll
                for (int m = 0; m < KMAX; ++m) {
                    row.s[m] = f32_sub(clamp0(t[m]), row.w[m]);
                    row.d[m] = f32_sub(f32_negmul(row.g[m], row.s[m]), f32_mul(hprr[m], f32_mul(row.s[m], row.s[m])));
                }
            } else {
#pragma unroll
                for (int m = 0; m < KMAX; ++m) {
                    row.g[m] = op_add(row.g[m], op_mul(sq, pq[m]));
                    t[m] = op_sub(row.w[m], greedy_div(row.g[m], row.den[m], row.rden[m]));
                }
#pragma unroll
                for (int m = 0; m < KMAX; ++m) {
                    row.s[m] = op_sub(clamp0(t[m]), row.w[m]);
                    row.d[m] = op_sub(op_mul(-row.g[m], row.s[m]), op_mul(hprr[m], op_mul(row.s[m], row.s[m])));
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < KMAX; ++m) {
                wnew[m] = (m == qm && lane == ql) ? op_add(wnew[m], sq) : wnew[m];
                {                                         // (lanes beyond k take P(q, c) = 0: G stays put; KMAX lane + m may lie beyond the padded row)
                    const T pq = (KMAX * lane + m < k) ? fetch(q, m) : (T)0;
                    row.g[m] = op_add(row.g[m], op_mul(sq, pq));
                    greedy_sd(row.w[m], row.g[m], row.prr[m], row.den[m], row.rden[m], row.s[m], row.d[m]);
                }
            }
        }
        row.template argmax<FULL>(k, km, lane, dq, q);
    }
#pragma unroll
    for (int m = 0; m < KMAX; ++m) {
        const int c = KMAX * lane + m;
        if (c < k) {
            T v = op_add(row.w[m], wnew[m]);
            v = (v < (T)0) ? (T)0 : v;      // projectnn!
            Wout.at(i, c) = v;
        }
    }
    return step;
}

// Rows take very different numbers of greedy steps (16384^2, k = 256, settled factors: mean 232, 99th percentile 343, maximum 577), and
// a launch of one row per wave ends with a few long rows stepping alone at the chain latency (~390 ns per step against the ~110 ns
// of eight waves sharing a SIMD).  So the launch is PERSISTENT: gridDim.x * 4 waves, as many as are resident at once; wave w starts
// with row w, and every further row comes from GREEDY_NQ shared counters (contiguous ranges of the remaining rows; a wave drains its
// home range, then helps with the others).  The counters live 128 bytes apart (atomics on one address serialise at ~30 ns each) and
// are zeroed by greedy_pinit_reduce_kernel, which always precedes this launch.  Which wave sweeps which row does not change any
// result: rows are independent.
constexpr int GREEDY_NQ = 32, GREEDY_QSTRIDE = 32;   // counters, ints between two counters
__device__ __forceinline__ int64_t greedy_next_row(int *queue, int64_t first_dyn, int64_t nsamples, int home, int lane) {
    const int64_t per = (nsamples - first_dyn + GREEDY_NQ - 1) / GREEDY_NQ;   // rows per range (the last ones may be short or empty)
    for (;;) {
        // lanes 0 .. NQ-1 look at one counter each (plain loads: exhausted ranges cost no atomic)
        const int64_t lo_l = first_dyn + (int64_t)lane * per;
        const int64_t len_l = (lane < GREEDY_NQ) ? (nsamples - lo_l < per ? nsamples - lo_l : per) : 0;
        int cnt = 0x7fffffff;
        if (lane < GREEDY_NQ && len_l > 0) cnt = __hip_atomic_load(queue + lane * GREEDY_QSTRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned open = (unsigned)__builtin_amdgcn_ballot_w64((int64_t)cnt < len_l);
        if (open == 0u) return nsamples;                                        // everything is handed out
        const unsigned rot = (open >> home) | (open << ((GREEDY_NQ - home) & 31));   // the home range first, then the following ones
        const int pick = (home + __builtin_ctz(rot)) & (GREEDY_NQ - 1);
        int r = 0;
        if (lane == 0) r = atomicAdd(queue + pick * GREEDY_QSTRIDE, 1);
        r = __builtin_amdgcn_readfirstlane(r);
        const int64_t lo = first_dyn + (int64_t)pick * per;
        const int64_t len = nsamples - lo < per ? nsamples - lo : per;
        if (r < len) return lo + r;                                             // else: lost the race for the range's last rows; look again
    }
}

template <typename T, int KMAX, bool FULL>   // FULL: ceil(k / 64) == KMAX (two kernels rather than two forms in one: each stays within 64 registers)
__global__ __launch_bounds__(256) void greedy_sweep_kernel(SampleView<const T> Wold, SampleView<T> Wout, SampleView<const T> G,
                                                           const T *__restrict__ P, int64_t ldp, int64_t nsamples, int k, T lambda,
                                                           T epsT, const T *pinit, int *queue, long long *steps_total, const int *done) {
    NMFX_DONE_GUARD(done);
    __shared__ long long nsteps[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t first_dyn = (int64_t)gridDim.x * 4;
    const int home = (int)(blockIdx.x & (GREEDY_NQ - 1));
    // (uniform row base + a 32-bit lane offset: the saddr form of the load, no 64-bit vector address arithmetic per step)
    const unsigned ldp32 = (unsigned)ldp, lane32 = (unsigned)lane;
    auto fetch = [&](int q, int m) { const T *rowp = P + (unsigned)q * ldp32; return rowp[(unsigned)KMAX * lane32 + (unsigned)m]; };
    long long steps = 0;
    for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < nsamples;
         i = (first_dyn < nsamples) ? greedy_next_row(queue, first_dyn, nsamples, home, lane) : nsamples) {
        // an opaque copy of the lane number per row: nothing lane-dependent is hoisted out of the row loop (the 64-bit offsets of the
        // row's loads and stores, P(r, r), 1 / (eps + P(r, r)): ~40 registers across the launch, i.e. 5 waves per SIMD instead of 8, to
        // save ~60 instructions per row)
        int lane_r = lane;
        asm volatile("" : "+v"(lane_r));
        steps += greedy_sweep_row<T, KMAX, FULL>(Wold, Wout, G, P, ldp, i, k, lambda, epsT, pinit, lane_r, fetch);
    }
    // executed greedy steps (nmfx_result.inner_iters): one atomic per block, not per row (16384 atomics on one address are
    // serialised at ~30 ns each on this chip)
    if (lane == 0) nsteps[wave] = steps;
    __syncthreads();
    if (threadIdx.x == 0 && steps_total != nullptr) {
        const long long tot = nsteps[0] + nsteps[1] + nsteps[2] + nsteps[3];
        if (tot > 0) atomicAdd((unsigned long long *)steps_total, (unsigned long long)tot);
    }
}

// (Measured and dropped: the same sweep with the packed upper triangle of P -- 131.6 KB for K = 256 in Float32 -- resident in LDS,
// 16 rows per workgroup sharing it: bit-identical, and SLOWER, 10.2 against 7.8 ms per iteration at 16384^2, k = 256, and 0.73
// against 0.68 ms even with one row per CU, where only the dependency chain counts: the L2 round trip for the row of P is not
// what the chain is made of -- the arg-max reduction and the four IEEE divisions of every step are.)

// sum_i |x_i| partials (norm(W, 1), greedycd.jl:81-86): Float64 accumulation, one partial per block
template <typename T> __global__ void sumabs_kernel(const T *x, int64_t count, double *partial, const int *done) {
    NMFX_DONE_GUARD(done);
    __shared__ double sm[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const T v = x[i];
        s += (double)(v < (T)0 ? -v : v);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// extra[slot] = T(lambda * T(sum partial))   (GreedyCD's L1 regulariser terms, greedycd.jl:81-86)
template <typename T>
__global__ void finish_sumabs_kernel(const double *partial, int n, T lambda, double *extra, int slot, const int *done) {
    NMFX_DONE_GUARD(done);
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += partial[i];
    extra[slot] = (double)(T)(lambda * (T)s);
}

// ---------------------------------------------------------------------------------------------------------------------
// k > 1024: the components of a sample row no longer fit the register file.  Same sweeps, one wave (= one workgroup) per
// sample row, the row's component vectors in LDS (component r at slot r: lane r % 64 reads it, conflict-free).  Every
// expression, the per-lane accumulation order (slots ascending) and the butterfly / DPP reductions are those of the
// register kernels above, so for a k both forms can run the results are bit-identical (tests/test_gpu_cd.py forces this
// form at small k with NMFX_CD_LDS=1).  The reference's loops (src/coorddesc.jl:133-156, src/greedycd.jl:134-158) have no
// size limit; this form's is the 160 KiB of LDS: k <= 20480 (cd, f32), ~4400 (greedycd, f32), ~2800 (greedycd, f64).
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void cd_sweep_lds_kernel(SampleView<const T> Wold, SampleView<T> Wnew, SampleView<const T> Z,
                                                          const T *__restrict__ P, int64_t ldp, int64_t nsamples, int k, T l1, const int *done) {
    NMFX_DONE_GUARD(done);
    extern __shared__ __attribute__((aligned(16))) unsigned char cd_lds_raw[];
    const int lane = threadIdx.x, kp = (k + 63) / 64 * 64;
    T *w = reinterpret_cast<T *>(cd_lds_raw), *z = w + kp;
    const int64_t i = blockIdx.x;
    for (int c = lane; c < kp; c += 64) {
        w[c] = (c < k) ? Wold.at(i, c) : (T)0;
        z[c] = (c < k) ? (T)(Z.at(i, c) - l1) : (T)0;
    }
    __syncthreads();
    for (int t = 0; t < k; ++t) {
        T part = (T)0;
        for (int c = lane; c < kp; c += 64) part += ((c < k) ? P[(int64_t)t * ldp + c] : (T)0) * w[c];
        const T hess = P[(int64_t)t * ldp + t];
        const T grad = wave_sum(part) - z[t];
        T nw = w[t] - grad / hess;
        nw = (nw > (T)0) ? nw : ((nw != nw) ? nw : (T)0);
        __syncthreads();                       // every lane has read w[t]
        if (hess != (T)0 && lane == 0) w[t] = nw;
        __syncthreads();
    }
    for (int c = lane; c < k; c += 64) Wnew.at(i, c) = w[c];
}

// the row's state in LDS: w, g, s, d, prr, den, wnew (T) and rden (double), kp elements each
template <typename T> struct GreedyLds {
    T *w, *g, *s, *d, *prr, *den, *wn;
    double *rden;
    __device__ __forceinline__ GreedyLds(unsigned char *raw, int kp) {
        rden = reinterpret_cast<double *>(raw);
        w = reinterpret_cast<T *>(rden + kp);
        g = w + kp; s = g + kp; d = s + kp; prr = d + kp; den = prr + kp; wn = den + kp;
    }
    static size_t bytes(int kp) { return (size_t)kp * (sizeof(double) + 7 * sizeof(T)); }
};
// (best, q) = max / arg-max of D over the row (first index on ties), both wave-uniform; D(c) from `dget(c)`
template <typename T, typename F> __device__ __forceinline__ void greedy_argmax_lds(F dget, int k, int kp, int lane, T &best, int &q) {
    best = -INFINITY;
    for (int c = lane; c < kp; c += 64) {
        const T dv = dget(c);
        const bool take = (c < k) && (dv > best);
        best = take ? dv : best;
    }
    best = wave_max_uniform(best);
    q = 0x7fffffff;
    for (int c0 = 0; c0 < kp; c0 += 64) {
        const int c = c0 + lane;
        const unsigned long long hit = __builtin_amdgcn_ballot_w64((c < k) && (dget(c) == best));
        if (hit != 0ull) { q = c0 + (int)__builtin_ctzll(hit); break; }
    }
}
template <typename T> __device__ __forceinline__ void greedy_load_lds(GreedyLds<T> &r, const SampleView<const T> &W, const SampleView<const T> &G,
                                                                    const T *P, int64_t ldp, int64_t i, int k, int kp, int lane, T lambda, T epsT) {
    for (int c = lane; c < kp; c += 64) {
        const bool ok = c < k;
        const T wv = ok ? W.at(i, c) : (T)0;
        T gv = ok ? G.at(i, c) : (T)0;
        if (ok && lambda > (T)0) gv = op_add(gv, lambda);
        const T pv = ok ? P[(int64_t)c * ldp + c] : (T)1;
        const T dn = op_add(epsT, pv);
        const double rd = 1.0 / (double)dn;
        T sv, dv;
        greedy_sd(wv, gv, pv, dn, rd, sv, dv);
        r.w[c] = wv; r.g[c] = gv; r.prr[c] = pv; r.den[c] = dn; r.rden[c] = rd; r.s[c] = sv; r.d[c] = dv; r.wn[c] = (T)0;
    }
}
template <typename T>
__global__ __launch_bounds__(64) void greedy_pinit_lds_kernel(SampleView<const T> W, SampleView<const T> G, const T *__restrict__ P, int64_t ldp,
                                                              int64_t nsamples, int k, T lambda, T epsT, T *part, const int *done) {
    NMFX_DONE_GUARD(done);
    extern __shared__ __attribute__((aligned(16))) unsigned char cd_lds_raw[];
    const int lane = threadIdx.x, kp = (k + 63) / 64 * 64;
    GreedyLds<T> r(cd_lds_raw, kp);
    greedy_load_lds(r, W, G, P, ldp, (int64_t)blockIdx.x, k, kp, lane, lambda, epsT);
    __syncthreads();
    T best; int q;
    greedy_argmax_lds<T>([&](int c) { return r.d[c]; }, k, kp, lane, best, q);
    if (lane == 0) part[blockIdx.x] = (best > (T)-1) ? best : (T)-1;
}
template <typename T>
__global__ __launch_bounds__(64) void greedy_sweep_lds_kernel(SampleView<const T> Wold, SampleView<T> Wout, SampleView<const T> G,
                                                              const T *__restrict__ P, int64_t ldp, int64_t nsamples, int k, T lambda, T epsT,
                                                              const T *pinit, long long *steps_total, const int *done) {
    NMFX_DONE_GUARD(done);
    extern __shared__ __attribute__((aligned(16))) unsigned char cd_lds_raw[];
    const int lane = threadIdx.x, kp = (k + 63) / 64 * 64;
    const int64_t i = blockIdx.x;
    GreedyLds<T> r(cd_lds_raw, kp);
    greedy_load_lds(r, Wold, G, P, ldp, i, k, kp, lane, lambda, epsT);
    __syncthreads();
    const T thresh = op_mul((T)0.001, pinit[0]);
    T dq; int q;
    greedy_argmax_lds<T>([&](int c) { return r.d[c]; }, k, kp, lane, dq, q);
    const long long max_steps = (long long)k * k;
    long long step = 0;
    for (; step < max_steps; ++step) {
        if (dq < thresh) break;
        const T sq = r.s[q];                      // broadcast read
        __syncthreads();
        if (lane == 0) r.wn[q] = op_add(r.wn[q], sq);
        for (int c = lane; c < kp; c += 64) {     // lanes c in [k, kp) read P's zero padding: G stays put (K >= kp: P is K x K zero-padded)
            const T pq = P[(int64_t)q * ldp + c];
            const T gv = op_add(r.g[c], op_mul(sq, pq));
            r.g[c] = gv;
            T sv, dv;
            greedy_sd(r.w[c], gv, r.prr[c], r.den[c], r.rden[c], sv, dv);
            r.s[c] = sv; r.d[c] = dv;
        }
        __syncthreads();
        greedy_argmax_lds<T>([&](int c) { return r.d[c]; }, k, kp, lane, dq, q);
    }
    if (lane == 0 && steps_total != nullptr && step > 0) atomicAdd((unsigned long long *)steps_total, (unsigned long long)step);
    __syncthreads();
    for (int c = lane; c < k; c += 64) {
        T v = op_add(r.w[c], r.wn[c]);
        v = (v < (T)0) ? (T)0 : v;
        Wout.at(i, c) = v;
    }
}

}  // namespace nmfx
