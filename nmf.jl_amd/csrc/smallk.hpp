// smallk.hpp -- MultUpdate (MSE) for k <= 64 in Float32: a low-latency path, 4 launches per outer iteration instead of 12.
// update_wh!(::MultUpdMSE), src/multupd.jl:83-116, in Gram form like solver_impl.hpp.
//
// At k = 64 (BASELINE config 2: 4096 x 4096) the general path is launch / latency bound: its two p*n*k products run 16-way split-K
// (27 us each against a 13.7 us MFMA floor, then a slab reduction), and ten more launches of 6-9 us carry ~1 us of work each.
// Here one workgroup owns a 16-wide stripe of the output for the WHOLE contraction, so nothing is split and everything that
// follows the product happens in the same launch:
//   smallk_h_kernel : stripe = 16 columns of X.  num = W' X[:, stripe] (64 x 16) over all p rows; den = (W'W) H[:, stripe];
//                     H <- H .* max(0, num - lh) ./ (den + delta)  (:98-103); stop_condition's sums for the stripe (common.jl:100-104);
//                     the stripe's contribution to HH' (64 x 64, for the W side's denominator, :110).
//   smallk_w_kernel : stripe = 16 rows of X.  num = X[stripe, :] H' (16 x 64) over all n columns; den = W[stripe, :] (HH');
//                     the W update (:109-114), its stop_condition sums, the stripe's contribution to W'W (next iteration's :99).
// After each: one launch sums the stripes' Gram contributions (fixed order) and one finalises the statistics.
// Matrix cores: v_mfma_f32_16x16x4_f32 (16 x 16 output tiles: a 64 x 16 stripe is 4 of them, one per wave; the contraction is
// split once more inside the workgroup, 8 waves = 2 per SIMD, halves combined through LDS).
// Measured alternatives (scripts/kbench/mfma16_probe.hip gives the issue-rate yardstick: 34 cycles per MFMA per SIMD, i.e. 14 us
// for one stripe pass at 4096^2): fragments loaded straight from global memory with no LDS and no barrier -- each wave owning
// 1/8 of the contraction for the whole stripe, three register sets, loads two chunks ahead (the schedule had to be pinned with
// sched_barrier: the machine scheduler sinks the loads to their uses) -- 50 us per launch against 37 us for the LDS-staged form;
// the same loop with the loads removed still takes 32 us, so ~17 us of every launch are not the main loop at all (launch, the
// cold first stage, the Gram / stripe prologue, combine + epilogue + stripe Gram), and the staged form is within 25 % of what
// this decomposition can do.
// Rounding: the numerator is two accumulation chains over the contraction per half (the general path sums 16 split-K slabs), the
// Gram contributions are summed stripe by stripe: same arithmetic class, not the same bits as the general path.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "kernels.hpp"

namespace nmfx {

typedef float smallk_v4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ smallk_v4 smallk_mfma(float a, float b, smallk_v4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int SMALLK_THREADS = 512;
// ST = contraction rows (H side) / columns (W side) per stage = per workgroup barrier: 128 (16 MFMAs per wave between barriers;
// 64 left the two waves of a SIMD waiting at a barrier for about as long as they computed)
constexpr int SMALLK_ST = 128;
// LDS floats: staged operands (2 stages) | Gram | old-factor stripe | half-combine | new-factor stripe
template <int ST> constexpr int smallk_h_lds() { return 2 * 64 * (ST + 4) + 2 * 16 * (ST + 4) + 64 * 80 + 16 * 68 + 4 * 64 * 4 + 16 * 80; }
template <int ST> constexpr int smallk_w_lds() { return 2 * ST * 80 + 2 * ST * 16 + 64 * 80 + 64 * 16 + 4 * 64 * 4 + 16 * 80; }
constexpr int SMALLK_H_LDS = smallk_h_lds<SMALLK_ST>();
constexpr int SMALLK_W_LDS = smallk_w_lds<SMALLK_ST>();

// multiplicative update of one element (EpiMultUpdate::apply, gemm_mfma.hpp): max(zero(T), num - lambda) with Julia's NaN rule
__device__ __forceinline__ float smallk_update(float ov, float nu, float dn, float lambda, float delta) {
    float t = nu - lambda;
    t = (t > 0.0f) ? t : ((t != t) ? t : 0.0f);
    return ov * (t / (dn + delta));
}

// block -> stripe: blocks are dealt round-robin to the 8 XCDs; neighbouring stripes (which share 128-byte lines of X on the W
// side) go to the same XCD's L2
__device__ __forceinline__ int smallk_stripe() {
    const int nb = gridDim.x, b = blockIdx.x;
    return ((nb & 7) == 0) ? (b & 7) * (nb >> 3) + (b >> 3) : b;
}

// the part both kernels share once the new stripe sits in LDS as S[line][comp] (16 lines of 64 components, row stride 80):
// its Gram contribution S' S (64 x 64) -> slab; 16 output tiles over 8 waves
__device__ __forceinline__ void smallk_stripe_gram(const float *S, float *slab, int wave, int i, int kg) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int tt = 2 * wave + q, tr = tt >> 2, tc = tt & 3;
        smallk_v4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 4; ++m) g = smallk_mfma(S[(4 * m + kg) * 80 + 16 * tr + i], S[(4 * m + kg) * 80 + 16 * tc + i], g);
        *reinterpret_cast<smallk_v4 *>(slab + (16 * tc + i) * 64 + 16 * tr + 4 * kg) = g;
    }
}

// H side.  X: P x N (ld ldx), W: P x 64 (ld ldx), gramW: 64 x 64, Ho / Hn: 64 x N (ld 64).  grid = N / 16, 512 threads.
// Main loop: 64 rows of the contraction per stage, both operands staged through LDS in their natural layout (coalesced float4
// loads, 256 contiguous bytes per 16 threads), double-buffered, one barrier per stage; two register sets keep the global loads
// two stages ahead.  Wave (w, half): output tile w (components 16w ..), contraction rows 32 half .. of every stage; every
// fragment read is a conflict-free ds_read_b32 (row strides 68 / 80 / 16 floats put the 64 lanes on 64 different banks).
// PROBE (scripts/kbench/smallk_probe.hip only; 0 in the library): 1 = no global loads inside the loop, 2 = no MFMAs, 3 = no
// staging at all (neither the LDS stores nor the global loads), 4 = no fragment reads either (MFMAs on registers)
template <int ST, int PROBE = 0>
__global__ __launch_bounds__(SMALLK_THREADS) void smallk_h_kernel(const float *X, int64_t ldx, int64_t P, const float *W, const float *gramW, const float *Ho,
                                                                  float *Hn, float lambda, float delta, float *gram_slabs, double *stat_part,
                                                                  const int *done) {
    if (done && *done) return;
    constexpr int LD = ST + 4;           // LDS row stride: 4 mod 32 banks, like 68
    constexpr int C4 = ST / 4;           // float4 chunks per staged row
    constexpr int NW = 64 * C4 / SMALLK_THREADS, NX = (16 * C4 + SMALLK_THREADS - 1) / SMALLK_THREADS;   // chunks per thread and stage
    constexpr bool XALL = (16 * C4 >= SMALLK_THREADS);
    extern __shared__ __attribute__((aligned(16))) float smallk_lds[];
    float *Wc = smallk_lds;              // [2][64 comps][LD]   W(p, comp), p contiguous
    float *Xc = Wc + 2 * 64 * LD;        // [2][16 cols][LD]    X(p, col)
    float *Gs = Xc + 2 * 16 * LD;        // [64 a][80]          gramW(comp, a), comp contiguous
    float *Hs = Gs + 64 * 80;            // [16 cols][68]       Ho(a, col), a contiguous
    float *Cx = Hs + 16 * 68;            // [4 waves][64 lanes][4]
    float *Sn = Cx + 4 * 64 * 4;         // [16 cols][80]       Hn(comp, col), comp contiguous
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, w = wave & 3, half = wave >> 2, i = lane & 15, kg = lane >> 4;
    const int stripe = smallk_stripe();
    const int64_t c0 = (int64_t)stripe * 16;
    // staging: W stage = 64 x C4 float4 (NW per thread: chunk tid + 512 q -> component chunk / C4), X stage = 16 x C4 float4
    const bool xld = XALL || tid < 16 * C4;
    smallk_v4 rw[2][NW], rx[2][NX];
#pragma unroll
    for (int q = 0; q < NX; ++q) { rx[0][q] = smallk_v4{0.f, 0.f, 0.f, 0.f}; rx[1][q] = rx[0][q]; }
    // buffer loads: the chunk's byte offset inside a stage is loop invariant (one 32-bit VGPR per chunk), the stage's position goes into
    // the scalar base of the descriptor -- no 64-bit vector address arithmetic per load (as in gemm_mfma.hpp's TileLoader::load_buf)
    uint32_t offw[NW], offx[NX];
#pragma unroll
    for (int q = 0; q < NW; ++q) { const int c = tid + SMALLK_THREADS * q; offw[q] = (uint32_t)(((int64_t)(c / C4) * ldx + 4 * (c % C4)) * 4); }
#pragma unroll
    for (int q = 0; q < NX; ++q) { const int c = tid + SMALLK_THREADS * q; offx[q] = (uint32_t)(((int64_t)(c / C4) * ldx + 4 * (c % C4)) * 4); }
    typedef unsigned smallk_v4u __attribute__((ext_vector_type(4)));
    auto gload = [&](int set, int64_t p0) {
        const __amdgpu_buffer_rsrc_t dw = __builtin_amdgcn_make_buffer_rsrc((void *)(W + p0), 0, -1, 0x00020000);
        const __amdgpu_buffer_rsrc_t dx = __builtin_amdgcn_make_buffer_rsrc((void *)(X + c0 * ldx + p0), 0, -1, 0x00020000);
#pragma unroll
        for (int q = 0; q < NW; ++q) rw[set][q] = __builtin_bit_cast(smallk_v4, __builtin_amdgcn_raw_buffer_load_b128(dw, (int)offw[q], 0, 0));
#pragma unroll
        for (int q = 0; q < NX; ++q)
            if (xld) rx[set][q] = __builtin_bit_cast(smallk_v4, __builtin_amdgcn_raw_buffer_load_b128(dx, (int)offx[q], 0, 0));
    };
    auto lstore = [&](int set, int buf) {
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const int c = tid + SMALLK_THREADS * q;
            *reinterpret_cast<smallk_v4 *>(Wc + buf * 64 * LD + (c / C4) * LD + 4 * (c % C4)) = rw[set][q];
        }
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            const int c = tid + SMALLK_THREADS * q;
            if (xld) *reinterpret_cast<smallk_v4 *>(Xc + buf * 16 * LD + (c / C4) * LD + 4 * (c % C4)) = rx[set][q];
        }
    };
    const int T = (int)(P / ST);      // even: P is a multiple of 256
    gload(0, 0);
    // operands of the epilogue (independent of the main loop): gramW and the old stripe -- requested behind the first stage so
    // that the cold round trips overlap
    {
        const int a = tid >> 4, c4 = tid & 15;
        const smallk_v4 g0 = *reinterpret_cast<const smallk_v4 *>(gramW + a * 64 + 4 * c4);
        const smallk_v4 g1 = *reinterpret_cast<const smallk_v4 *>(gramW + (a + 32) * 64 + 4 * c4);
        smallk_v4 h0 = {0.f, 0.f, 0.f, 0.f};
        if (tid < 256) h0 = *reinterpret_cast<const smallk_v4 *>(Ho + (c0 + (tid >> 4)) * 64 + 4 * (tid & 15));
        lstore(0, 0);
        gload(0, ST);
        gload(1, (int64_t)((T > 2) ? 2 : T - 1) * ST);      // (unconditional: see the step below)
        *reinterpret_cast<smallk_v4 *>(Gs + a * 80 + 4 * c4) = g0;
        *reinterpret_cast<smallk_v4 *>(Gs + (a + 32) * 80 + 4 * c4) = g1;
        if (tid < 256) *reinterpret_cast<smallk_v4 *>(Hs + (tid >> 4) * 68 + 4 * (tid & 15)) = h0;
    }
    __syncthreads();
    smallk_v4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    auto step = [&](int t, auto SET) {
        constexpr int set = decltype(SET)::value;      // = t & 1: stage t+1 waits in register set t & 1
        const int buf = t & 1;
        // Both operands are contraction-contiguous in LDS: ONE ds_read_b128 per operand feeds four MFMAs -- lane (i, kg) holds rows
        // 16 j + 4 kg + {0..3} of its line, MFMA m of group j contracts rows 16 j + {m, 4 + m, 8 + m, 12 + m} (any partition of the
        // rows into fours is a valid order of the sum).  ALL fragments of the stage are requested before the first MFMA: written as
        // `acc = mfma(wa[..], xb[..], acc)` in a loop the compiler issued each pair of reads behind the previous pair's MFMAs and waited
        // lgkmcnt(0) in front of every pair -- an LDS round trip exposed 8 times per stage (MFMA pipe 48 % busy).
        const float *wa = Wc + buf * 64 * LD + (16 * w + i) * LD + (ST / 2) * half + 4 * kg;
        const float *xb = Xc + buf * 16 * LD + i * LD + (ST / 2) * half + 4 * kg;
        constexpr int NJ = ST / 32;
        smallk_v4 fa[NJ], fb[NJ];
        if constexpr (PROBE != 4) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                fa[j] = *reinterpret_cast<const smallk_v4 *>(wa + 16 * j);
                fb[j] = *reinterpret_cast<const smallk_v4 *>(xb + 16 * j);
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if constexpr (PROBE == 2) {
                acc0 += fa[j] * fb[j];
            } else if constexpr (PROBE == 4) {
                acc0 = smallk_mfma(acc1[1], acc1[2], acc0);
                acc1 = smallk_mfma(acc0[1], acc0[2], acc1);
                acc0 = smallk_mfma(acc1[0], acc1[3], acc0);
                acc1 = smallk_mfma(acc0[0], acc0[3], acc1);
            } else {
                acc0 = smallk_mfma(fa[j][0], fb[j][0], acc0);
                acc1 = smallk_mfma(fa[j][1], fb[j][1], acc1);
                acc0 = smallk_mfma(fa[j][2], fb[j][2], acc0);
                acc1 = smallk_mfma(fa[j][3], fb[j][3], acc1);
            }
        }
        // Staging is UNCONDITIONAL (the last steps re-load the final stage and store into a buffer nobody reads): with `if (t + 3 < T)`
        // around the loads the compiler could not count them at the store below and waited vmcnt(0) there -- for the loads issued one
        // step ago as well as for the ones it needs, i.e. the two-stages-ahead prefetch was one stage deep.
        if constexpr (PROBE < 3) lstore(set, buf ^ 1);
        if constexpr (PROBE == 0 || PROBE == 2) gload(set, (int64_t)((t + 3 < T) ? t + 3 : T - 1) * ST);
        __syncthreads();
    };
    for (int t = 0; t < T; t += 2) {
        step(t, std::integral_constant<int, 0>{});
        step(t + 1, std::integral_constant<int, 1>{});
    }
    smallk_v4 num = acc0 + acc1;
    if (half) *reinterpret_cast<smallk_v4 *>(Cx + (w * 64 + lane) * 4) = num;
    __syncthreads();
    if (!half) {
        num += *reinterpret_cast<const smallk_v4 *>(Cx + (w * 64 + lane) * 4);
        // den = (W'W)[comps of this wave, :] * Ho[:, stripe]
        smallk_v4 den = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 16; ++m) den = smallk_mfma(Gs[(4 * m + kg) * 80 + 16 * w + i], Hs[i * 68 + 4 * m + kg], den);
        // lane: column c0 + i, components 16w + 4kg + r
        smallk_v4 nv;
        double dv[4], sv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ov = Hs[i * 68 + 16 * w + 4 * kg + r];
            nv[r] = smallk_update(ov, num[r], den[r], lambda, delta);
            const float d = nv[r] - ov, sp = nv[r] + ov;
            dv[r] = (double)(float)(d * d);
            sv[r] = (double)(float)(sp * sp);
        }
        *reinterpret_cast<smallk_v4 *>(Hn + (c0 + i) * 64 + 16 * w + 4 * kg) = nv;
        *reinterpret_cast<smallk_v4 *>(Sn + i * 80 + 16 * w + 4 * kg) = nv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) { dv[r] += __shfl_xor(dv[r], off, 64); sv[r] += __shfl_xor(sv[r], off, 64); }
        }
        if (i == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int comp = 16 * w + 4 * kg + r;
                stat_part[((int64_t)stripe * 64 + comp) * 2] = dv[r];
                stat_part[((int64_t)stripe * 64 + comp) * 2 + 1] = sv[r];
            }
        }
    }
    __syncthreads();
    smallk_stripe_gram(Sn, gram_slabs + (int64_t)stripe * 4096, wave, i, kg);
}

// W side.  gramH: 64 x 64, Wo / Wn: P x 64 (ld ldx), H: 64 x N (ld 64).  grid = P / 16, 512 threads.
// Same structure with ST COLUMNS per stage (H staged component-contiguous, X row-contiguous).
template <int ST>
__global__ __launch_bounds__(SMALLK_THREADS) void smallk_w_kernel(const float *X, int64_t ldx, int64_t N, const float *H, const float *gramH, const float *Wo,
                                                                  float *Wn, float lambda, float delta, float *gram_slabs, double *stat_part,
                                                                  const int *done) {
    if (done && *done) return;
    extern __shared__ __attribute__((aligned(16))) float smallk_lds[];
    constexpr int NH = ST * 16 / SMALLK_THREADS, NX = (ST * 4 + SMALLK_THREADS - 1) / SMALLK_THREADS;   // float4 chunks per thread and stage
    constexpr bool XALL = (ST * 4 >= SMALLK_THREADS);
    float *Hc = smallk_lds;              // [2][ST cols][80]    H(comp, col), comp contiguous
    float *Xc = Hc + 2 * ST * 80;        // [2][ST cols][16]    X(row, col), row contiguous
    float *Gs = Xc + 2 * ST * 16;        // [64 a][80]          gramH(comp, a)
    float *Ws = Gs + 64 * 80;            // [64 a][16 rows]     Wo(row, a), row contiguous
    float *Cx = Ws + 64 * 16;            // [4][64][4]
    float *Sn = Cx + 4 * 64 * 4;         // [16 rows][80]       Wn(row, comp), comp contiguous
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, w = wave & 3, half = wave >> 2, i = lane & 15, kg = lane >> 4;
    const int stripe = smallk_stripe();
    const int64_t r0 = (int64_t)stripe * 16;
    // staging: H stage = ST cols x 64 comps = 16 ST float4 (NH per thread: chunk tid + 512 q -> column chunk / 16), X stage = ST cols x 16 rows
    const bool xld = XALL || tid < ST * 4;
    smallk_v4 rh[2][NH], rx[2][NX];      // two stages ahead, as in smallk_h_kernel
#pragma unroll
    for (int q = 0; q < NX; ++q) { rx[0][q] = smallk_v4{0.f, 0.f, 0.f, 0.f}; rx[1][q] = rx[0][q]; }
    uint32_t offh[NH], offx[NX];           // buffer loads with loop-invariant lane offsets (see smallk_h_kernel)
#pragma unroll
    for (int q = 0; q < NH; ++q) { const int c = tid + SMALLK_THREADS * q; offh[q] = (uint32_t)(((c >> 4) * 64 + 4 * (c & 15)) * 4); }
#pragma unroll
    for (int q = 0; q < NX; ++q) { const int c = tid + SMALLK_THREADS * q; offx[q] = (uint32_t)((4 * (c & 3) + (int64_t)(c >> 2) * ldx) * 4); }
    auto gload = [&](int set, int64_t j0) {
        const __amdgpu_buffer_rsrc_t dh = __builtin_amdgcn_make_buffer_rsrc((void *)(H + j0 * 64), 0, -1, 0x00020000);
        const __amdgpu_buffer_rsrc_t dx = __builtin_amdgcn_make_buffer_rsrc((void *)(X + r0 + j0 * ldx), 0, -1, 0x00020000);
#pragma unroll
        for (int q = 0; q < NH; ++q) rh[set][q] = __builtin_bit_cast(smallk_v4, __builtin_amdgcn_raw_buffer_load_b128(dh, (int)offh[q], 0, 0));
#pragma unroll
        for (int q = 0; q < NX; ++q)
            if (xld) rx[set][q] = __builtin_bit_cast(smallk_v4, __builtin_amdgcn_raw_buffer_load_b128(dx, (int)offx[q], 0, 0));
    };
    auto lstore = [&](int set, int buf) {
#pragma unroll
        for (int q = 0; q < NH; ++q) {
            const int c = tid + SMALLK_THREADS * q;
            *reinterpret_cast<smallk_v4 *>(Hc + buf * ST * 80 + (c >> 4) * 80 + 4 * (c & 15)) = rh[set][q];
        }
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            const int c = tid + SMALLK_THREADS * q;
            if (xld) *reinterpret_cast<smallk_v4 *>(Xc + buf * ST * 16 + (c >> 2) * 16 + 4 * (c & 3)) = rx[set][q];
        }
    };
    const int T = (int)(N / ST);      // even: N is a multiple of 256
    gload(0, 0);
    {
        const int a = tid >> 4, c4 = tid & 15;
        const smallk_v4 g0 = *reinterpret_cast<const smallk_v4 *>(gramH + a * 64 + 4 * c4);
        const smallk_v4 g1 = *reinterpret_cast<const smallk_v4 *>(gramH + (a + 32) * 64 + 4 * c4);
        smallk_v4 w0 = {0.f, 0.f, 0.f, 0.f};
        if (tid < 256) w0 = *reinterpret_cast<const smallk_v4 *>(Wo + r0 + 4 * (tid & 3) + (int64_t)(tid >> 2) * ldx);
        lstore(0, 0);
        gload(0, ST);
        gload(1, (int64_t)((T > 2) ? 2 : T - 1) * ST);      // (unconditional: see the step below)
        *reinterpret_cast<smallk_v4 *>(Gs + a * 80 + 4 * c4) = g0;
        *reinterpret_cast<smallk_v4 *>(Gs + (a + 32) * 80 + 4 * c4) = g1;
        if (tid < 256) *reinterpret_cast<smallk_v4 *>(Ws + (tid >> 2) * 16 + 4 * (tid & 3)) = w0;
    }
    __syncthreads();
    smallk_v4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    auto step = [&](int t, auto SET) {
        constexpr int set = decltype(SET)::value;
        const int buf = t & 1;
        const float *ha = Hc + buf * ST * 80 + ((ST / 2) * half + kg) * 80 + 16 * w + i;
        const float *xb = Xc + buf * ST * 16 + ((ST / 2) * half + kg) * 16 + i;
        // all fragments of the stage first, then the MFMAs; staging unconditional (see smallk_h_kernel)
        constexpr int NM = ST / 8;
        float fa[NM], fb[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            fa[m] = ha[4 * m * 80];
            fb[m] = xb[4 * m * 16];
        }
#pragma unroll
        for (int m = 0; m < NM; m += 2) {
            acc0 = smallk_mfma(fa[m], fb[m], acc0);
            acc1 = smallk_mfma(fa[m + 1], fb[m + 1], acc1);
        }
        lstore(set, buf ^ 1);
        gload(set, (int64_t)((t + 3 < T) ? t + 3 : T - 1) * ST);
        __syncthreads();
    };
    for (int t = 0; t < T; t += 2) {
        step(t, std::integral_constant<int, 0>{});
        step(t + 1, std::integral_constant<int, 1>{});
    }
    smallk_v4 num = acc0 + acc1;
    if (half) *reinterpret_cast<smallk_v4 *>(Cx + (w * 64 + lane) * 4) = num;
    __syncthreads();
    if (!half) {
        num += *reinterpret_cast<const smallk_v4 *>(Cx + (w * 64 + lane) * 4);
        // den(comp, row) = sum_a (HH')(comp, a) Wo(row, a)
        smallk_v4 den = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 16; ++m) den = smallk_mfma(Gs[(4 * m + kg) * 80 + 16 * w + i], Ws[(4 * m + kg) * 16 + i], den);
        // lane: row r0 + i, components 16w + 4kg + r
        smallk_v4 nv;
        double dv[4], sv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int comp = 16 * w + 4 * kg + r;
            const float ov = Ws[comp * 16 + i];
            nv[r] = smallk_update(ov, num[r], den[r], lambda, delta);
            const float d = nv[r] - ov, sp = nv[r] + ov;
            dv[r] = (double)(float)(d * d);
            sv[r] = (double)(float)(sp * sp);
            Wn[r0 + i + (int64_t)comp * ldx] = nv[r];
        }
        *reinterpret_cast<smallk_v4 *>(Sn + i * 80 + 16 * w + 4 * kg) = nv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) { dv[r] += __shfl_xor(dv[r], off, 64); sv[r] += __shfl_xor(sv[r], off, 64); }
        }
        if (i == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int comp = 16 * w + 4 * kg + r;
                stat_part[((int64_t)stripe * 64 + comp) * 2] = dv[r];
                stat_part[((int64_t)stripe * 64 + comp) * 2 + 1] = sv[r];
            }
        }
    }
    __syncthreads();
    smallk_stripe_gram(Sn, gram_slabs + (int64_t)stripe * 4096, wave, i, kg);
}

// After a stripe kernel, ONE launch: blocks [0, 128) sum the stripes' Gram contributions, blocks [128, 128 + 32) sum the statistics
// partials (128 values, a wave per value as in finalize_partials_kernel).
constexpr int SMALLK_GRAM_BLOCKS = 128;
static __global__ __launch_bounds__(256) void smallk_finish_kernel(float *gram, const float *slabs, int nstripes, const double *stat_part, double *stat_out,
                                                           const int *done, Ctrl *ctrl = nullptr, const double *hstat = nullptr, int k = 0, float tol = 0.f,
                                                           long long t = 0, unsigned *ticket = nullptr) {
    if (done && *done) return;
    if (blockIdx.x < SMALLK_GRAM_BLOCKS) {
        // 32 Gram elements (one 128-byte line of every slab) per block: thread = (float4 e4 of the line, slab lane sl of 32); a wave-load
        // covers 8 whole lines; the 32 slab-lane partials are added in lane order (fixed order)
        __shared__ smallk_v4 sm[32][8];
        const int e4 = threadIdx.x & 7, sl = threadIdx.x >> 3;
        const int64_t i = (int64_t)blockIdx.x * 32 + 4 * e4;
        smallk_v4 acc = {0.f, 0.f, 0.f, 0.f};
        // (eight loads in flight, then the adds in stripe order: written as `acc += load` in a loop every load waited for the add in front of
        // it -- 8 dependent round trips at 256 stripes, most of this launch's 6 us)
        for (int k0 = sl; k0 < nstripes; k0 += 32 * 8) {
            smallk_v4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = (k0 + 32 * u < nstripes) ? k0 + 32 * u : k0;
                v[u] = *reinterpret_cast<const smallk_v4 *>(slabs + (int64_t)k * 4096 + i);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k0 + 32 * u < nstripes) acc += v[u];
        }
        sm[sl][e4] = acc;
        __syncthreads();
        if (sl == 0) {
            smallk_v4 t = sm[0][e4];
#pragma unroll
            for (int q = 1; q < 32; ++q) t += sm[q][e4];
            *reinterpret_cast<smallk_v4 *>(gram + i) = t;
        }
    } else {
        const int e = (blockIdx.x - SMALLK_GRAM_BLOCKS) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        double s = 0.0;
        for (int c0 = lane; c0 < nstripes; c0 += 64 * 4) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = (c0 + 64 * u < nstripes) ? c0 + 64 * u : c0;
                v[u] = stat_part[(int64_t)c * 128 + e];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (c0 + 64 * u < nstripes) s += v[u];
        }
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) stat_out[e] = s;
        if (ctrl != nullptr) {
            // W side, not tracking: the stop rule of this iteration (check_kernel's work) runs in the LAST of the 32 statistics blocks
            // to finish -- 32 device-scope increments, not a launch (4.9 us per iteration at 4096^2, k = 64).  Agent-scope release by
            // every block, acquire by the last one (cdna_hip_programming.md); the counter only ever grows, 32 per launch.
            __shared__ int last;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                const unsigned prev = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = ((prev & 31u) == 31u) ? 1 : 0;
                if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            if (last) check_body<float>(ctrl, stat_out, hstat, k, tol, t, nullptr);
        }
    }
}

}  // namespace nmfx
