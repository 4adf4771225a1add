// gemm_bf16x3.hpp -- optional mixed-precision form of the two p*n*k products: the fp32
// operands are split on the fly into bf16 pairs a = hi + lo (hi = bf16(a), lo = bf16(a - hi): 16 mantissa bits kept) and
// the product is formed as  hi*hi + hi*lo + lo*hi  with three v_mfma_f32_32x32x16_bf16 per tile and k-step, accumulated
// in fp32 -- 16x the per-instruction rate of the fp32 MFMA for 3x the instructions.  Measured on the C3 shapes: 440 us per
// launch (310 TFLOP/s fp32-equivalent, 2.2x the fp32 kernel) with a maximum relative error of 1.7e-6 against fp64 -- smaller
// than a sequential fp32 loop's 4.7e-6, because the dropped lo*lo terms (2^-16 relative, random sign) average out over the
// 16384-term sums while the accumulation is the same fp32.  It is NOT the default: nmfx_opts.precision = NMFX_PREC_BF16X3
// selects it (f32 contexts, k >= 65), and the objective trajectory stays within the 1e-5 tolerance (tests/test_gpu_bf16x3.py).
// Same conventions as gemm_mfma.hpp: D(r, c) = sum_k A(r, k) B(c, k), element (r, c) of split-K slab s at D[s*stride + c + r*ld].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_mfma.hpp"

namespace nmfx {
namespace bf16x3 {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BR = 128, BC = 128, BK = 32, NT = 256;
constexpr int PLANE = BR * BK * 2;          // bytes of one bf16 plane of one operand tile (8 KiB)
constexpr int OPER = 2 * PLANE;             // hi + lo
constexpr int STAGE = 2 * OPER;             // A + B  (32 KiB)

// LDS image of one bf16 plane: [chunk (8 k = 16 B)][row][16 B]; the row slot is permuted so that the fragment reads
// (32 consecutive rows, one chunk), the KCONTIG writes (one row, 4 chunks x 2 halves per 8 lanes) and the KSTRIDED
// micro-tile writes (rows 4 apart) are all (nearly) conflict-free:  slot = row ^ ((row >> 4) & 3) ^ (chunk << 2)
__device__ __forceinline__ int lds_off(int row, int chunk) { return chunk * (BR * 16) + ((row ^ ((row >> 4) & 3) ^ (chunk << 2)) << 4); }

// split 4 consecutive-k fp32 values into bf16 hi / lo quads
__device__ __forceinline__ void split4(const f32x4 v, bf16x4 &hi, bf16x4 &lo) {
    hi = __builtin_convertvector(v, bf16x4);
    const f32x4 hf = __builtin_convertvector(hi, f32x4);
    lo = __builtin_convertvector(v - hf, bf16x4);
}

__device__ __forceinline__ void load_tile(f32x4 (&r)[4], const float *base, int64_t ld, int64_t row0, int64_t k0, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = tid + NT * i;
        const int row = s >> 3, cpos = s & 7;          // 8 quads (32 k) per row
        r[i] = *reinterpret_cast<const f32x4 *>(base + (row0 + row) * ld + k0 + cpos * 4);
    }
}
__device__ __forceinline__ void store_tile(const f32x4 (&r)[4], char *oper, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = tid + NT * i;
        const int row = s >> 3, cpos = s & 7;
        bf16x4 hi, lo;
        split4(r[i], hi, lo);
        const int off = lds_off(row, cpos >> 1) + ((cpos & 1) << 3);
        *reinterpret_cast<bf16x4 *>(oper + off) = hi;
        *reinterpret_cast<bf16x4 *>(oper + PLANE + off) = lo;
    }
}
// KSTRIDED operand (element (row, k) at base[k*ld + row]): a thread owns one 4 x 4 micro-tile (4 consecutive rows x 4
// consecutive k): four 16-byte loads, transposed in registers at store time
__device__ __forceinline__ void load_tile_ks(f32x4 (&r)[4], const float *base, int64_t ld, int64_t row0, int64_t k0, int tid) {
    const int kq = tid >> 5, r4 = tid & 31;
#pragma unroll
    for (int ek = 0; ek < 4; ++ek) r[ek] = *reinterpret_cast<const f32x4 *>(base + (k0 + kq * 4 + ek) * ld + row0 + r4 * 4);
}
__device__ __forceinline__ void store_tile_ks(const f32x4 (&r)[4], char *oper, int tid) {
    const int kq = tid >> 5, r4 = tid & 31;
#pragma unroll
    for (int er = 0; er < 4; ++er) {
        const f32x4 v = {r[0][er], r[1][er], r[2][er], r[3][er]};
        bf16x4 hi, lo;
        split4(v, hi, lo);
        const int off = lds_off(r4 * 4 + er, kq >> 1) + ((kq & 1) << 3);
        *reinterpret_cast<bf16x4 *>(oper + off) = hi;
        *reinterpret_cast<bf16x4 *>(oper + PLANE + off) = lo;
    }
}
__device__ __forceinline__ bf16x8 frag(const char *plane, int rt, int ks, int lane) {
    const int r = rt + (lane & 31), c = 2 * ks + (lane >> 5);
    return *reinterpret_cast<const bf16x8 *>(plane + lds_off(r, c));
}

// Operand layouts per side (LA / LB): 0 = KCONTIG (A(r,k) at A[r*lda + k]), 1 = KSTRIDED (A(r,k) at A[k*lda + r]).
// Epi is one of the epilogue functors of gemm_mfma.hpp (same accumulator layout as the fp32 32x32 MFMA: 16 registers per
// tile, col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)), so the fused ratio / objective / store epilogues are shared.
struct Args {
    const float *A, *B;
    int64_t lda, ldb;
    int tiles_r, tiles_c, splits, kchunk, c_fastest, group;
    const int *done;
};

template <int LA, int LB, typename Epi>
__global__ __launch_bounds__(NT) void gemm_bf16x3_kernel(Args g, Epi epi) {
    if (g.done != nullptr && *reinterpret_cast<const volatile int *>(g.done) != 0) return;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int tiles = g.tiles_r * g.tiles_c;
    int split = bid / tiles;
    const int trem = bid % tiles;
    int tr, tc;
    if (g.group > 1) {   // 2-D super-tiles (see gemm_mfma_kernel)
        const int G = g.group, per = G * G;
        const int st = trem / per, in = trem % per;
        const int sr = st / (g.tiles_c / G), sc = st % (g.tiles_c / G);
        tr = sr * G + in / G;
        tc = sc * G + in % G;
    } else if (g.c_fastest) { tc = trem % g.tiles_c; tr = trem / g.tiles_c; }
    else { tr = trem % g.tiles_r; tc = trem / g.tiles_r; }
    tr = __builtin_amdgcn_readfirstlane(tr);
    tc = __builtin_amdgcn_readfirstlane(tc);
    split = __builtin_amdgcn_readfirstlane(split);
    const int64_t r0 = (int64_t)tr * BR, c0 = (int64_t)tc * BC, kbeg = (int64_t)split * g.kchunk;
    const int nk = g.kchunk / BK;
    TileCtx tctx{tr, tc, wr, wc, lane, tid, NT, (int)blockIdx.x, r0, c0, 4 * (lane >> 5), lane & 31, r0 + wr * 64, c0 + wc * 64};
    epi.setup(split, tctx);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    f32x4 ra[4], rb[4];
    auto lda_ = [&](int64_t k0) { if constexpr (LA) load_tile_ks(ra, g.A, g.lda, r0, k0, tid); else load_tile(ra, g.A, g.lda, r0, k0, tid); };
    auto ldb_ = [&](int64_t k0) { if constexpr (LB) load_tile_ks(rb, g.B, g.ldb, c0, k0, tid); else load_tile(rb, g.B, g.ldb, c0, k0, tid); };
    auto sta_ = [&](char *oper) { if constexpr (LA) store_tile_ks(ra, oper, tid); else store_tile(ra, oper, tid); };
    auto stb_ = [&](char *oper) { if constexpr (LB) store_tile_ks(rb, oper, tid); else store_tile(rb, oper, tid); };
    lda_(kbeg); ldb_(kbeg);
    sta_(smem); stb_(smem + OPER);
    if (nk > 1) { lda_(kbeg + BK); ldb_(kbeg + BK); }
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        const char *a_s = smem + (t & 1) * STAGE, *b_s = a_s + OPER;
        char *a_n = smem + ((t & 1) ^ 1) * STAGE, *b_n = a_n + OPER;
        if (t + 1 < nk) { sta_(a_n); stb_(b_n); }
        if (t + 2 < nk) { lda_(kbeg + (int64_t)(t + 2) * BK); ldb_(kbeg + (int64_t)(t + 2) * BK); }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { ah[i] = frag(a_s, wr * 64 + i * 32, ks, lane); al[i] = frag(a_s + PLANE, wr * 64 + i * 32, ks, lane); }
#pragma unroll
            for (int j = 0; j < 2; ++j) { bh[j] = frag(b_s, wc * 64 + j * 32, ks, lane); bl[j] = frag(b_s + PLANE, wc * 64 + j * 32, ks, lane); }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    // epilogue: per row of MFMA tiles, request the inputs, then compute and store (gemm_mfma.hpp conventions)
    epi.begin();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        typename Epi::Pre pre[2][16];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) pre[j][reg] = epi.prefetch(i * 32 + (reg & 3) + 8 * (reg >> 2), j * 32);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) epi.apply(i * 32 + (reg & 3) + 8 * (reg >> 2), j * 32, acc[i][j][reg], j, pre[j][reg]);
        __builtin_amdgcn_sched_barrier(0);
    }
    epi.template finish<32, 2, 2, 2>(reinterpret_cast<double *>(smem), tctx);
}

}  // namespace bf16x3
}  // namespace nmfx
