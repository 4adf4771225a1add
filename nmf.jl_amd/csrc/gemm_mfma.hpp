// gemm_mfma.hpp -- LDS-tiled MFMA GEMM for gfx950 with pluggable epilogues.
//
// Every GEMM of the NMF hot path goes through this one kernel template:
//   W'X, W'W, (W'W)H           (src/multupd.jl:98-99, src/projals.jl:92-93, src/alspgrad.jl:63-67,124)
//   XH', HH', W(HH')           (src/multupd.jl:109-110, src/projals.jl:100-101, src/alspgrad.jl:218-222,280)
//   WH (never materialised; feeds the objective / ratio epilogues)
//                              (src/multupd.jl:104,115,172-174; src/multupd.jl:81,148)
//
// The kernel computes  D(r, c) = sum_k A(r, k) * B(c, k)  for an R x C output whose
// memory address is  c + r*ld  (c is the CONTIGUOUS index of the column-major Julia
// matrix, r the strided one).  MFMA lanes run along c, so every epilogue load/store is a
// 128-byte coalesced segment per 32 (f32) / 16 (f64) lanes.
//
// All operands are column-major Julia matrices, so an operand is either
//   KCONTIG  : element (row, k) at base[row*ld + k]   (contraction index contiguous)
//   KSTRIDED : element (row, k) at base[k*ld + row]   (row index contiguous)
// Dimensions are pre-padded by the host (multiples of the block tile / BK) so the
// main loop has no bounds checks; padding is zero and algebraically inert.
//
// f32 uses v_mfma_f32_32x32x2_f32 (exact f32), f64 uses v_mfma_f64_16x16x4_f64.
// Within a k-group of 8 the lane's k-slot ks and step q map to k = 8g + VEC*ks + q for BOTH
// operands, so one 16-byte LDS read feeds VEC consecutive MFMAs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include <type_traits>
#include <utility>

namespace nmfx {

enum : int { KCONTIG = 0, KSTRIDED = 1 };

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f64x2 = __attribute__((ext_vector_type(2))) double;
using f64x4 = __attribute__((ext_vector_type(4))) double;

template <typename T> struct Mfma;
template <> struct Mfma<float> {
    static constexpr int MT = 32, KS = 2, VEC = 4, NACC = 16, BK = 32;
    using acc_t = f32x16;
    using vec_t = f32x4;
    // D = A(32 x 2) * B(2 x 32) + C ; lane l supplies A[l&31][l>>5] and B[l>>5][l&31]
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
};
template <> struct Mfma<double> {
    static constexpr int MT = 16, KS = 4, VEC = 2, NACC = 4, BK = 16;
    using acc_t = f64x4;
    using vec_t = f64x2;
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
};

template <typename T> struct GemmArgs {
    const T *A;             // rows <-> r (strided output index)
    const T *B;             // rows <-> c (contiguous output index)
    int64_t lda, ldb;
    int tiles_r, tiles_c;   // grid of block tiles
    int splits;             // split-K factor
    int kchunk;             // contraction length handled by one split (multiple of BK)
    int c_fastest;          // 1: consecutive blocks walk c-tiles first (they share the A tile)
    const int *done;        // device stop flag: kernels of iterations past the stop are no-ops
    // Optional second segment of an operand (fuses the k x k Gram into the big GEMM launch):
    // rows r >= r_split come from A2 (ld lda2), rows c >= c_split from B2 (ld ldb2).  Splits are tile-aligned.
    const T *A2 = nullptr;
    const T *B2 = nullptr;
    int64_t lda2 = 0, ldb2 = 0;
    int64_t r_split = INT64_MAX, c_split = INT64_MAX;
    int group = 1;          // >1: super-tile rasterisation (see the block -> tile mapping)
    const T *a_aux = nullptr, *b_aux = nullptr;   // operand computed on the fly as max(z - alpha*g, 0) - z (see TileLoader::load)
    const double *alpha_ptr = nullptr;            // device-resident step size (PgState::alpha)
    int tail_tiles = 0;     // extra output tiles along the slow tile direction, done as a balanced second segment
    int tail_nkt = 0;       // k-tiles of a tail tile (= Kdim / BK)
    int tail_per = 0;       // k-tiles of one tail piece (one piece per block)
    int tail_main = 0;      // > 0: the grid is `tail_main` blocks SHORT of tiles * splits; the missing (tile, split) items -- the
                            // last ones -- are dealt out as tail pieces instead (a grid that must leave some CUs free)
    int prio = 0;           // wave priority pattern of the launch (see the kernel): 0 none
    // B operand BLOCKED along the contraction (KCONTIG): contraction rows [q * b_blk_k, (q + 1) * b_blk_k) live in a matrix of their own
    // with leading dimension ldb = b_blk_k, block q starting b_blk_stride elements after block q - 1 (the row-sharded W as the
    // all-gather delivers it: rank q's rows x k, contiguous).  A block's k-range must not straddle blocks (b_blk_k % kchunk == 0).
    int64_t b_blk_k = 0, b_blk_stride = 0;
    // Operands that live in one of several equally spaced buffer SETS whose index is only known on the device (the projected-gradient
    // sub-solver's rotating (Z, G) sets, pgrad.hpp): set = *sel, read once at kernel start; the operand's base moves by set * stride
    // elements (a stride of 0 leaves an operand alone).
    const int *sel = nullptr;
    int64_t a_sel = 0, b_sel = 0, aaux_sel = 0, baux_sel = 0;
};

// Epilogues that stage their output through the block's LDS (free behind the main loop) declare `static constexpr bool USES_LDS`: the
// kernel hands them the buffer before the first apply() and calls row_done(i) behind every row of MFMA tiles (EpiStorePeer).
template <typename E, typename = void> struct epi_uses_lds : std::false_type {};
template <typename E> struct epi_uses_lds<E, std::void_t<decltype(E::USES_LDS)>> : std::true_type {};

// what an epilogue may need to know about the block / wave it runs in
struct TileCtx {
    int tr, tc, wr, wc, lane, tid, nthreads, bid;
    int64_t r0, c0;     // block tile origin
    int rl, cl;         // the lane's row / column inside an MFMA output tile
    int64_t rw0, cw0;   // wave tile origin (wave-uniform)
};

// Epilogue addressing.  Element (r, c) of the wave's output tile splits into a WAVE-UNIFORM part (wave tile origin
// (rw0, cw0), MFMA tile and accumulator register: compile-time (ro, co) relative to the origin) and a LANE part (rl, cl)
// that is the same for every element a lane owns.  Every array an epilogue touches gets a buffer descriptor based at
// the wave tile origin; element (ro, co) is then  buffer_load/store v, v_lane_off, s[desc], s_off offen  with ONE
// shared 32-bit lane-offset VGPR and the row/column offset in an SGPR: no per-element address lives in VGPRs.  (The
// plain  base[c + r*ld]  form kept a 64-bit VGPR pointer per accumulator row and prefetch set -- 30-60 VGPRs that put
// the ratio / update / gradient kernels at one wave per SIMD -- plus 64-bit VALU address arithmetic per element.)
// Offsets are 32-bit and relative to the wave tile: 256 * ld * sizeof(T) must stay below 2^32 (checked by the host).
using rsrc_t = __amdgpu_buffer_rsrc_t;
template <typename T> __device__ __forceinline__ rsrc_t tile_rsrc(const T *base, int64_t ld, const TileCtx &t) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)(base + (t.cw0 + t.rw0 * ld)), 0, -1, 0x00020000);
}
template <typename T> __device__ __forceinline__ T buf_ld(rsrc_t rs, uint32_t voff, uint32_t soff) {
    if constexpr (sizeof(T) == 4) return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)soff, 0));
    else return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, (int)soff, 0));
}
template <typename T> __device__ __forceinline__ void buf_st(rsrc_t rs, uint32_t voff, uint32_t soff, T v) {
    if constexpr (sizeof(T) == 4) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, (int)voff, (int)soff, 0);
    } else {
        typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_t, v), rs, (int)voff, (int)soff, 0);
    }
}
// lane part and row pitch of an array with leading dimension ld, in bytes
template <typename T> struct LaneAddr {
    uint32_t lb, pitch;
    __device__ __forceinline__ void init(const TileCtx &t, int64_t ld) {
        lb = (uint32_t)(((int64_t)t.cl + (int64_t)t.rl * ld) * (int64_t)sizeof(T));
        pitch = (uint32_t)(ld * (int64_t)sizeof(T));
    }
    __device__ __forceinline__ uint32_t soff(int ro, int co) const { return (uint32_t)ro * pitch + (uint32_t)co * (uint32_t)sizeof(T); }
};

// XOR swizzle of the 16-byte chunk position inside a KCONTIG LDS row (8 chunks/row):
// conflict-free ds_read_b128 for 16 rows distinct mod 16 (MI355X LDS: 64 banks x 4 B).
__device__ __forceinline__ int swz8(int row) { return (row >> 1) & 7; }
// Row permutation of the KSTRIDED micro-tile image (16-byte slot index inside one k-quad): makes BOTH the
// ds_write_b128 of a micro-tile row (lanes 4 rows apart: 64-byte stride) and the ds_read_b128 of a fragment
// (lanes on consecutive rows) conflict-free.
__device__ __forceinline__ int swzrow(int r) { return r ^ ((r >> 3) & 3); }

// KSTRIDED tiles use the transposing micro-tile image when every thread owns whole VEC x VEC micro-tiles; small
// tiles (fewer chunks per thread than VEC) keep the plain [k][row] image.
template <typename T, int ROWS, int NTHREADS> constexpr bool kstrided_micro() {
    // f32 only: with f64 (2 x 2 micro-tiles, 8-byte fragment reads already pair up) the plain image measured faster
    return sizeof(T) == 4 && ((ROWS * Mfma<T>::BK / Mfma<T>::VEC / NTHREADS) % Mfma<T>::VEC) == 0 &&
           (ROWS * Mfma<T>::BK / Mfma<T>::VEC / NTHREADS) > 0;
}

template <typename T, int LAYOUT, int ROWS, int NTHREADS> struct TileLoader {
    using M = Mfma<T>;
    static constexpr int VEC = M::VEC, BK = M::BK;
    static constexpr int CHUNKS = ROWS * BK / VEC;
    static constexpr int PER_THREAD = CHUNKS / NTHREADS;
    static_assert(CHUNKS % NTHREADS == 0, "tile chunks must divide evenly among threads");
    using vec_t = typename M::vec_t;

    // global -> registers
    // aux != nullptr: the operand is not read but COMPUTED on the fly from two arrays with identical addressing,
    //   d = max(z - alpha*g, 0) - z   (z from `base`, g from `aux`)
    // i.e. the projected-gradient trial step D = Zn - Z of src/alspgrad.jl:142-147, which therefore never exists in memory.
    // (AUX is a compile-time flag: a run-time test here would split the software-pipelined main loop into basic
    // blocks and void its issue-order template -- measured 142 -> 133 TF/s on the big GEMM.)
    // AM = 2: the operand is the SUM of two arrays with identical addressing (two split-K slabs: the combine launch folded into the
    // consumer's loader; z + g in this order, like reduce_slabs_kernel)
    template <int AM = 0>
    static __device__ __forceinline__ void load(vec_t (&r)[PER_THREAD], const T *base, int64_t ld,
                                                int64_t row0, int64_t k0, int tid, const T *aux = nullptr, T alpha = (T)0) {
        if constexpr (LAYOUT == KCONTIG) {
#pragma unroll
            for (int i = 0; i < PER_THREAD; ++i) {
                const int s = tid + NTHREADS * i;
                constexpr int CPR = BK / VEC;   // 8 chunks per row
                const int row = s / CPR, cpos = s % CPR;
                const int c = cpos ^ swz8(row);
                const T *p = base + (row0 + row) * ld + k0 + c * VEC;
                r[i] = *reinterpret_cast<const vec_t *>(p);
                if constexpr (AM != 0) xform<AM>(r[i], *reinterpret_cast<const vec_t *>(aux + (p - base)), alpha);
            }
        } else {
            // KSTRIDED: a thread owns VEC x VEC micro-tiles (VEC rows x VEC consecutive k): VEC global loads of 16 bytes
            // (rows contiguous), transposed in registers (free: only the register naming changes), so that the LDS
            // image is [k/VEC][row][VEC] and one ds_read_b128 yields a lane's VEC consecutive k -- the same fragment
            // read as the KCONTIG image (the [k][row] image needed 4x as many LDS reads: 134.6 vs 142 TF/s).
          if constexpr (kstrided_micro<T, ROWS, NTHREADS>()) {
            constexpr int RPV = ROWS / VEC;   // micro-tile columns (groups of VEC rows)
#pragma unroll
            for (int m = 0; m < PER_THREAD / VEC; ++m) {
                const int mt = tid + NTHREADS * m;
                const int kq = mt / RPV, r4 = mt % RPV;
                // r[m*VEC + ek] = raw chunk of k-row ek; the transpose happens in store(), one k-tile later, when the
                // data has long arrived (transposing here would put an s_waitcnt vmcnt right behind the loads)
#pragma unroll
                for (int ek = 0; ek < VEC; ++ek) {
                    const T *p = base + (k0 + kq * VEC + ek) * ld + row0 + r4 * VEC;
                    r[m * VEC + ek] = *reinterpret_cast<const vec_t *>(p);
                    if constexpr (AM != 0) xform<AM>(r[m * VEC + ek], *reinterpret_cast<const vec_t *>(aux + (p - base)), alpha);
                }
            }
          } else {
#pragma unroll
            for (int i = 0; i < PER_THREAD; ++i) {
                const int s = tid + NTHREADS * i;
                constexpr int CPK = ROWS / VEC;
                const int kk = s / CPK, r4 = s % CPK;
                const T *p = base + (k0 + kk) * ld + row0 + r4 * VEC;
                r[i] = *reinterpret_cast<const vec_t *>(p);
                if constexpr (AM != 0) xform<AM>(r[i], *reinterpret_cast<const vec_t *>(aux + (p - base)), alpha);
            }
          }
        }
    }
    // Buffer-load form of load() (plain operands only): the 32-bit byte offsets of a thread's chunks relative to the tile's first
    // element at k-tile 0 are loop invariant (offsets()), the k-tile's position goes into the scalar base of a buffer descriptor, so a
    // k-tile costs no vector address arithmetic (the pointer form spends two 64-bit VALU adds per load and a 64-bit VGPR pair).
    static __device__ __forceinline__ void offsets(uint32_t (&off)[PER_THREAD], int64_t ld, int tid) {
        if constexpr (LAYOUT == KCONTIG) {
#pragma unroll
            for (int i = 0; i < PER_THREAD; ++i) {
                const int s = tid + NTHREADS * i;
                constexpr int CPR = BK / VEC;
                const int row = s / CPR, cpos = s % CPR, c = cpos ^ swz8(row);
                off[i] = (uint32_t)(((int64_t)row * ld + c * VEC) * (int64_t)sizeof(T));
            }
        } else if constexpr (kstrided_micro<T, ROWS, NTHREADS>()) {
            constexpr int RPV = ROWS / VEC;
#pragma unroll
            for (int m = 0; m < PER_THREAD / VEC; ++m) {
                const int mt = tid + NTHREADS * m, kq = mt / RPV, r4 = mt % RPV;
#pragma unroll
                for (int ek = 0; ek < VEC; ++ek) off[m * VEC + ek] = (uint32_t)(((int64_t)(kq * VEC + ek) * ld + r4 * VEC) * (int64_t)sizeof(T));
            }
        } else {
#pragma unroll
            for (int i = 0; i < PER_THREAD; ++i) {
                const int s = tid + NTHREADS * i;
                constexpr int CPK = ROWS / VEC;
                const int kk = s / CPK, r4 = s % CPK;
                off[i] = (uint32_t)(((int64_t)kk * ld + r4 * VEC) * (int64_t)sizeof(T));
            }
        }
    }
    // tile base of k-tile k0 (what the descriptor points at)
    static __device__ __forceinline__ const T *tile_base(const T *base, int64_t ld, int64_t row0, int64_t k0) {
        if constexpr (LAYOUT == KCONTIG) return base + row0 * ld + k0;
        else return base + k0 * ld + row0;
    }
    // AUX: the operand computed on the fly from two arrays with identical addressing (see load()): `auxtb` = the tile base in the
    // second array
    template <int AM = 0>
    static __device__ __forceinline__ void load_buf(vec_t (&r)[PER_THREAD], const T *tb, const uint32_t (&off)[PER_THREAD], const T *auxtb = nullptr,
                                                    T alpha = (T)0) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)tb, 0, -1, 0x00020000);
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) r[i] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off[i], 0, 0));
        if constexpr (AM != 0) {
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)auxtb, 0, -1, 0x00020000);
#pragma unroll
            for (int i = 0; i < PER_THREAD; ++i) xform<AM>(r[i], __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off[i], 0, 0)), alpha);
        }
    }
    template <int AM = 1> static __device__ __forceinline__ void xform(vec_t &z, const vec_t &gv, T alpha) {
        if constexpr (AM == 2) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) z[q] = z[q] + gv[q];
        } else {
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const T zz = z[q];
                T v = zz - alpha * gv[q];
                v = (v > (T)0) ? v : ((v != v) ? v : (T)0);
                z[q] = v - zz;
            }
        }
    }
    // registers -> LDS.  KCONTIG: linear image, chunk s at byte 16*s.  KSTRIDED: micro-tile (kq, r4), row er at
    // 16-byte slot kq*ROWS + r4*VEC + er.
    static __device__ __forceinline__ void store(const vec_t (&r)[PER_THREAD], T *lds, int tid) {
        if constexpr (LAYOUT == KCONTIG || !kstrided_micro<T, ROWS, NTHREADS>()) {
#pragma unroll
            for (int i = 0; i < PER_THREAD; ++i) {
                const int s = tid + NTHREADS * i;
                *reinterpret_cast<vec_t *>(lds + s * VEC) = r[i];
            }
        } else {
            constexpr int RPV = ROWS / VEC;
#pragma unroll
            for (int m = 0; m < PER_THREAD / VEC; ++m) {
                const int mt = tid + NTHREADS * m;
                const int kq = mt / RPV, r4 = mt % RPV;
#pragma unroll
                for (int er = 0; er < VEC; ++er) {
                    vec_t t;
#pragma unroll
                    for (int ek = 0; ek < VEC; ++ek) t[ek] = r[m * VEC + ek][er];
                    *reinterpret_cast<vec_t *>(lds + (kq * ROWS + swzrow(r4 * VEC + er)) * VEC) = t;
                }
            }
        }
    }
};

// Read the VEC operand values of k-group g for the MFMA row-tile starting at tile row rt.
template <typename T, int LAYOUT, int ROWS, int NTHREADS>
__device__ __forceinline__ void read_frag(T (&out)[Mfma<T>::VEC], const T *lds, int rt, int g, int lane) {
    using M = Mfma<T>;
    const int r = rt + (lane % M::MT);
    const int ks = lane / M::MT;
    if constexpr (LAYOUT == KCONTIG) {
        const int c = (g * M::KS + ks) ^ swz8(r);
        const typename M::vec_t v =
            *reinterpret_cast<const typename M::vec_t *>(lds + (r * (M::BK / M::VEC) + c) * M::VEC);
#pragma unroll
        for (int q = 0; q < M::VEC; ++q) out[q] = v[q];
    } else if constexpr (kstrided_micro<T, ROWS, NTHREADS>()) {
        const typename M::vec_t v =
            *reinterpret_cast<const typename M::vec_t *>(lds + ((g * M::KS + ks) * ROWS + swzrow(r)) * M::VEC);
#pragma unroll
        for (int q = 0; q < M::VEC; ++q) out[q] = v[q];
    } else {
#pragma unroll
        for (int q = 0; q < M::VEC; ++q) out[q] = lds[(g * 8 + ks * M::VEC + q) * ROWS + r];
    }
}

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N-1>{})
template <int N, typename F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F &&f) {
    static_for_impl<N>(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}
// N x { 2 MFMA, 1 instruction of class MASK } in issue order (LLVM sched_group_barrier)
template <int MASK, int N> __device__ __forceinline__ void sched_pairs() {
    if constexpr (N > 0) {
        __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
        __builtin_amdgcn_sched_group_barrier(MASK, 1, 0);
        sched_pairs<MASK, N - 1>();
    }
}

// AUX: 0 plain operands; 1 / 2: operand A / B is the projected-gradient trial step computed in the loader; 3 / 4: operand A / B is the
// sum of two arrays (a_aux / b_aux = the second one: two split-K slabs summed on the way in; measured on ProjectedALS's solve products in
// round 4 and not used there -- the short grid's tail pieces still need their own combine launch, which costs what the fold saves).
// BUF: 1 = operand loads as buffer loads with loop-invariant lane offsets and the k-tile's position in the scalar descriptor
// (TileLoader::load_buf) instead of per-load 64-bit pointer arithmetic on the vector unit: -2.7 ... -4.5 % on every big product
// (scripts/kbench/gemm_bench.hip, A/B interleaved in one process: WtX 1169 -> 1138 us, XHt 1161 -> 1116 us on that box; the 8-rank
// shard shapes 170 -> 165 and 155 -> 148 us; Float64 573 -> 563 / 552 us).  The library launches every GEMM with BUF = 1.
template <typename T, int LA, int LB, int BR, int BC, int WGR, int WGC, typename Epi, int AUX = 0, int BUF = 0>
// half-size and smaller tiles ask for >= 2 waves per SIMD (they exist to overlap one block's prologue / epilogue with another
// block's MFMAs); without the hint the fused f64 epilogues land a few registers above the 256-register budget of two waves
__global__ __launch_bounds__(WGR *WGC * 64, (BR * BC <= 64 * 128) ? 2 : 1) void gemm_mfma_kernel(GemmArgs<T> g, Epi epi) {
    using M = Mfma<T>;
    constexpr int NT = WGR * WGC * 64;
    constexpr int BK = M::BK, MT = M::MT;
    constexpr int WTR = BR / WGR, WTC = BC / WGC;
    constexpr int TR = WTR / MT, TC = WTC / MT;
    static_assert(WTR % MT == 0 && WTC % MT == 0, "wave tile must be a multiple of the MFMA tile");
    using LoadA = TileLoader<T, LA, BR, NT>;
    using LoadB = TileLoader<T, LB, BC, NT>;

    if (g.done != nullptr && *reinterpret_cast<const volatile int *>(g.done) != 0) return;
    if (g.sel != nullptr) {
        const int64_t set = *g.sel;
        g.A += set * g.a_sel;
        g.B += set * g.b_sel;
        if (g.a_aux != nullptr) g.a_aux += set * g.aaux_sel;
        if (g.b_aux != nullptr) g.b_aux += set * g.baux_sel;
    }
    // Two blocks share a CU for the whole launch (one wave of each per SIMD).  With equal priorities the SIMD's issue arbiter
    // alternates between them; raising ONE of the two lets that wave run as if it were alone (an in-order wave keeps the matrix
    // pipe ~85-90 % busy by itself) while the other fills its gaps.  1: second half of the grid; 2: odd blocks; 3: every block.
    if (g.prio == 1) { if (blockIdx.x >= (gridDim.x >> 1)) __builtin_amdgcn_s_setprio(2); }
    else if (g.prio == 2) { if (blockIdx.x & 1) __builtin_amdgcn_s_setprio(2); }
    else if (g.prio == 3) __builtin_amdgcn_s_setprio(2);
    else if (g.prio == 4) { if ((blockIdx.x >> 3) & 1) __builtin_amdgcn_s_setprio(2); }
    T xalpha = (T)0;
    if constexpr (AUX == 1 || AUX == 2) xalpha = (T)*g.alpha_ptr;
    constexpr int AMA = (AUX == 1) ? 1 : ((AUX == 3) ? 2 : 0), AMB = (AUX == 2) ? 1 : ((AUX == 4) ? 2 : 0);
    __shared__ __attribute__((aligned(16))) T smem[2 * (BR + BC) * BK];
    constexpr int STAGE = (BR + BC) * BK;   // stage s: A tile at smem + s*STAGE, B tile right behind it

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (SGPR)
    const int wr = wave / WGC, wc = wave % WGC;

    // block -> (tile_r, tile_c, split).  Blocks are dealt round-robin to the 8 XCDs (block b runs
    // on XCD b % 8, each with a private L2), so re-index first: the blocks of one XCD get a
    // contiguous range of logical ids and neighbours that share an operand tile share an L2.
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int tiles = g.tiles_r * g.tiles_c;
    // Work decomposition.
    //   main part : block = (tile, split) over the tiles_r x tiles_c MAIN tiles; split s handles k-tiles
    //               [s*kchunk/BK, (s+1)*kchunk/BK) and writes slab s.  All blocks of a split walk k in lockstep, so the
    //               operand slices they share (W for every X tile, the X tile for both c-tiles) are L2 hits.
    //   tail part : `tail_tiles` extra output tiles appended along the slow tile direction (the k x k Gram riding in the
    //               big GEMM launch).  Their (tile, k-tile) units are dealt out evenly, `tail_per` k-tiles per block, as
    //               a second short segment of every block: each block does kchunk/BK + tail_per k-tiles, so a grid of
    //               exactly 2 blocks per CU stays balanced (a plain extra-tiles grid would put the 8 Gram blocks into a
    //               second wave: +50 %), and the main part keeps its k-alignment (a full stream-K split de-phases the
    //               blocks in k and re-fetches W from the fabric for every tile row: 2.9 GB instead of 1.3 GB per launch).
    //               Tail piece p of a tail tile writes tail slab p (epilogue: begin(-1 - p)).
    const int nkt = g.kchunk / BK;
    TileCtx tctx{0, 0, wr, wc, lane, tid, NT, (int)blockIdx.x, 0, 0,
                 (sizeof(T) == 4) ? 4 * (lane >> 5) : (lane >> 4), lane % MT, 0, 0};
    for (int phase = 0; phase < 2; ++phase) {
        int trem, split, kt0, nk;
        bool tail = false;
        if (phase == 0) {
            split = bid / tiles;
            trem = bid % tiles;
            kt0 = split * nkt;
            nk = nkt;
        } else {
            if (g.tail_main > 0) {
                // items nblk .. nblk + tail_main - 1 of the main decomposition, tail_per k-tiles per block; piece p of every
                // item goes to tail slab p, at the item's own tile coordinates
                const int ppt = (nkt + g.tail_per - 1) / g.tail_per;
                const int piece = bid % ppt, item = nblk + bid / ppt;
                if (bid / ppt >= g.tail_main) break;
                const int pk = piece * g.tail_per;
                nk = (nkt - pk < g.tail_per) ? (nkt - pk) : g.tail_per;
                if (nk <= 0) break;
                kt0 = (item / tiles) * nkt + pk;
                trem = item % tiles;
                split = -1 - piece;
                tail = true;
            } else {
            if (g.tail_tiles == 0) break;
            const int inner = g.c_fastest ? g.tiles_c : g.tiles_r;     // tail tiles extend the slow (outer) direction
            const int pieces_per_tile = (g.tail_nkt + g.tail_per - 1) / g.tail_per;
            const int piece = bid % pieces_per_tile, ttile = bid / pieces_per_tile;
            if (ttile >= g.tail_tiles * inner) break;
            kt0 = piece * g.tail_per;
            nk = (g.tail_nkt - kt0 < g.tail_per) ? (g.tail_nkt - kt0) : g.tail_per;
            if (nk <= 0) break;
            split = -1 - piece;
            trem = tiles + ttile;          // tile ids continue past the main tiles along the outer direction
            tail = true;
            }
        }
        int tr, tc;
        if (g.group > 1) {
            // 2-D super-tiles of group x group block tiles (both tile counts are multiples of `group`): the blocks of
            // one super-tile are consecutive logical ids, i.e. they run on ONE XCD at about the same time, so each
            // operand tile is fetched into that XCD's 4 MiB L2 once and re-used `group` times (outputs that are large
            // in both dimensions -- W*H for the objective / the ratio pass -- otherwise re-stream one operand per tile row).
            const int G = g.group, per = G * G;
            const int st = trem / per, in = trem % per;
            const int sr = st / (g.tiles_c / G), sc = st % (g.tiles_c / G);
            tr = sr * G + in / G;
            tc = sc * G + in % G;
        } else if (g.c_fastest) { tc = trem % g.tiles_c; tr = trem / g.tiles_c; }
        else                    { tr = trem % g.tiles_r; tc = trem / g.tiles_r; }
        // integer divisions by run-time values are expanded on the vector unit: move the (wave-uniform) results back to
        // SGPRs so that tile origins, operand base pointers, the k-loop trip count and all epilogue addresses stay scalar
        tr = __builtin_amdgcn_readfirstlane(tr);
        tc = __builtin_amdgcn_readfirstlane(tc);
        split = __builtin_amdgcn_readfirstlane(split);
        kt0 = __builtin_amdgcn_readfirstlane(kt0);
        nk = __builtin_amdgcn_readfirstlane(nk);
        const int64_t r0 = (int64_t)tr * BR, c0 = (int64_t)tc * BC;
        const int64_t kbeg = (int64_t)kt0 * BK;
        const T *Ab = g.A, *Bb = g.B;
        int64_t lda = g.lda, ldb = g.ldb, ra0 = r0, cb0 = c0;
        if (r0 >= g.r_split) { Ab = g.A2; lda = g.lda2; ra0 = r0 - g.r_split; }
        if (c0 >= g.c_split) { Bb = g.B2; ldb = g.ldb2; cb0 = c0 - g.c_split; }
        else if (g.b_blk_k > 0) { const int64_t q = kbeg / g.b_blk_k; Bb += q * (g.b_blk_stride - g.b_blk_k); }
        tctx.tr = tr; tctx.tc = tc; tctx.r0 = r0; tctx.c0 = c0;
        tctx.rw0 = r0 + wr * WTR; tctx.cw0 = c0 + wc * WTC;

        typename M::acc_t acc[TR][TC];
    #pragma unroll
        for (int i = 0; i < TR; ++i)
    #pragma unroll
            for (int j = 0; j < TC; ++j)
    #pragma unroll
                for (int r = 0; r < M::NACC; ++r) acc[i][j][r] = (T)0;

        // Software pipeline (one barrier per k-tile, no MFMA-free phase besides it):
        //   registers hold k-tile t+1 while the MFMAs of tile t run out of LDS stage t&1;
        //   first half of the k-groups : registers -> LDS stage (t+1)&1   (ds_write interleaved with MFMAs)
        //   second half                : global   -> registers, tile t+2  (loads interleaved with MFMAs)
        // Stage (t+1)&1 was last read during tile t-1, i.e. before the barrier that ended iteration t-1.
        // Epilogue inputs of the first row of MFMA tiles are requested BEFORE the main loop (epilogues that read
        // memory: the X tile of the ratio / objective passes, numerator and old factor of the multiplicative update), so
        // their HBM round trip runs under the MFMAs; the following rows are requested one row ahead of their use.
        constexpr bool EARLY = Epi::EARLY && sizeof(T) == 4;
        typename Epi::Pre pre[2][TC][M::NACC];
        // MFMA C/D layout: f32 32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5);
        // f64 16x16: col = lane&15, row = (lane>>4) + 4*reg.  col <-> c (contiguous), row <-> r.
        auto reg_row = [](int reg) { return (sizeof(T) == 4) ? ((reg & 3) + 8 * (reg >> 2)) : 4 * reg; };
        epi.setup(split, tctx);
        auto prefetch_row = [&](auto IC, auto SC) {
            constexpr int i = decltype(IC)::value, set = decltype(SC)::value;
#pragma unroll
            for (int j = 0; j < TC; ++j)
#pragma unroll
                for (int reg = 0; reg < M::NACC; ++reg) pre[set][j][reg] = epi.prefetch(i * MT + reg_row(reg), j * MT);
        };
        if constexpr (EARLY) prefetch_row(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});

        typename M::vec_t ra[LoadA::PER_THREAD], rb[LoadB::PER_THREAD];
        uint32_t offA[LoadA::PER_THREAD], offB[LoadB::PER_THREAD];
        if constexpr (BUF != 0) { LoadA::offsets(offA, lda, tid); LoadB::offsets(offB, ldb, tid); }
        auto loadA = [&](int64_t kk) {
            if constexpr (BUF != 0) {
                const T *tb = LoadA::tile_base(Ab, lda, ra0, kk);
                LoadA::template load_buf<AMA>(ra, tb, offA, (AMA != 0) ? g.a_aux + (tb - Ab) : nullptr, xalpha);
            } else LoadA::template load<AMA>(ra, Ab, lda, ra0, kk, tid, g.a_aux, xalpha);
        };
        auto loadB = [&](int64_t kk) {
            if constexpr (BUF != 0) {
                const T *tb = LoadB::tile_base(Bb, ldb, cb0, kk);
                LoadB::template load_buf<AMB>(rb, tb, offB, (AMB != 0) ? g.b_aux + (tb - Bb) : nullptr, xalpha);
            } else LoadB::template load<AMB>(rb, Bb, ldb, cb0, kk, tid, g.b_aux, xalpha);
        };
        loadA(kbeg);
        loadB(kbeg);
        LoadA::store(ra, smem, tid);
        LoadB::store(rb, smem + BR * BK, tid);
        {
            const int64_t k1 = kbeg + (int64_t)((nk > 1) ? 1 : 0) * BK;
            loadA(k1);
            loadB(k1);
        }
        __syncthreads();

        constexpr int NG = BK / 8;   // k-groups per tile
        static_assert(NG % 2 == 0, "fragment double buffer assumes an even number of k-groups");
        // Operand fragments are double-buffered in registers and fetched ONE k-group ahead: group kg's MFMAs run on set
        // kg&1 while the ds_reads of group kg+1 (issued behind the first MFMA pair) fill the other set, so an LDS round
        // trip (~130-200 cycles for four b128 reads) is covered by ~14 MFMAs instead of 2.  The first group of the next
        // tile is fetched right behind the barrier that publishes it, in front of the last group's MFMAs.
        // f64 (FDB = false) keeps ONE set: its wave tile needs 2-3x the fragment registers (16x16 MFMA tiles), and both the
        // second set and the carried first-group set were measured as net losses there -- without them the 128 x 128 f64
        // kernel fits two waves per SIMD (123 + 128 registers) and runs 2-4 % faster; every k-group reads its own
        // fragments at the start of the group.
        constexpr bool FDB = (sizeof(T) == 4);
        constexpr bool CARRY = FDB;   // first-group fragments of the next tile fetched behind the barrier
        T af[2][TR][M::VEC], bf[2][TC][M::VEC];
#pragma unroll
        for (int i = 0; i < TR; ++i) if constexpr (CARRY) read_frag<T, LA, BR, NT>(af[FDB ? 0 : 1][i], smem, wr * WTR + i * MT, 0, lane);
#pragma unroll
        for (int j = 0; j < TC; ++j) if constexpr (CARRY) read_frag<T, LB, BC, NT>(bf[FDB ? 0 : 1][j], smem + BR * BK, wc * WTC + j * MT, 0, lane);
        // BUF == 2: the k-loop unrolled by two with the LDS stage a compile-time constant (stage offsets become instruction
        // immediates instead of one vector add per LDS access)
        auto ktile = [&](int t, int cur) {
            const T *a_s = smem + cur * STAGE, *b_s = a_s + BR * BK;
            T *a_n = smem + (cur ^ 1) * STAGE, *b_n = a_n + BR * BK;
            const int tn = (t + 2 < nk) ? t + 2 : nk - 1;   // clamped: the last iterations re-load the final tile (never used)
            const int64_t kn = kbeg + (int64_t)tn * BK;
            static_for<NG>([&](auto KGC) {
                constexpr int kg = decltype(KGC)::value;
                constexpr bool last = (kg == NG - 1);
                constexpr int fc = FDB ? (kg & 1) : ((kg == 0 && CARRY) ? 1 : 0);   // set the MFMAs of this group read
                constexpr int fn = FDB ? (fc ^ 1) : 1;                    // set the next group's / next tile's first fragments go to
                constexpr bool stA = (kg == 0), stB = (kg == (NG > 2 ? 1 : 0));
                // (issuing the global loads one k-group earlier, right behind their ds_write, measured no faster: the loads
                // are not what the loop waits for)
                constexpr bool ldA = (kg == NG / 2), ldB = (kg == (NG > 2 ? NG / 2 + 1 : NG / 2));
                if constexpr (FDB && !last) {
#pragma unroll
                    for (int i = 0; i < TR; ++i) read_frag<T, LA, BR, NT>(af[fn][i], a_s, wr * WTR + i * MT, kg + 1, lane);
#pragma unroll
                    for (int j = 0; j < TC; ++j) read_frag<T, LB, BC, NT>(bf[fn][j], b_s, wc * WTC + j * MT, kg + 1, lane);
                }
                if constexpr (!FDB && (kg != 0 || !CARRY)) {
#pragma unroll
                    for (int i = 0; i < TR; ++i) read_frag<T, LA, BR, NT>(af[0][i], a_s, wr * WTR + i * MT, kg, lane);
#pragma unroll
                    for (int j = 0; j < TC; ++j) read_frag<T, LB, BC, NT>(bf[0][j], b_s, wc * WTC + j * MT, kg, lane);
                }
                if constexpr (stA) LoadA::store(ra, a_n, tid);
                if constexpr (stB) LoadB::store(rb, b_n, tid);
                if constexpr (last) {
                    // every LDS read of tile t and every LDS write of tile t+1 by this wave is issued: publish tile t+1,
                    // then fetch its first fragments while the MFMAs below still run on tile t's registers
                    __syncthreads();
#pragma unroll
                    for (int i = 0; i < TR; ++i) if constexpr (CARRY) read_frag<T, LA, BR, NT>(af[fn][i], a_n, wr * WTR + i * MT, 0, lane);
#pragma unroll
                    for (int j = 0; j < TC; ++j) if constexpr (CARRY) read_frag<T, LB, BC, NT>(bf[fn][j], b_n, wc * WTC + j * MT, 0, lane);
                }
                if constexpr (ldA) loadA(kn);
                if constexpr (ldB) loadB(kn);
#pragma unroll
                for (int q = 0; q < M::VEC; ++q)
#pragma unroll
                    for (int i = 0; i < TR; ++i)
#pragma unroll
                        for (int j = 0; j < TC; ++j) acc[i][j] = M::mma(af[fc][i][q], bf[fc][j][q], acc[i][j]);
                // Issue-order template for this k-group (LLVM sched_group_barrier; masks: MFMA 0x8, VMEM read 0x20,
                // DS read 0x100, DS write 0x200): one MFMA pair, the next group's fragment reads, then the staging traffic
                // of this group spread one instruction per MFMA pair, so neither the LDS writes nor the global loads open
                // an MFMA-free window.
                constexpr int NMFMA = M::VEC * TR * TC;
                constexpr int NFRAG = ((LA == KCONTIG || kstrided_micro<T, BR, NT>()) ? TR : TR * M::VEC) +
                                      ((LB == KCONTIG || kstrided_micro<T, BC, NT>()) ? TC : TC * M::VEC);
                constexpr int NW0 = (stA ? LoadA::PER_THREAD : 0) + (stB ? LoadB::PER_THREAD : 0);
                constexpr int NL0 = (ldA ? LoadA::PER_THREAD : 0) + (ldB ? LoadB::PER_THREAD : 0);
                constexpr int LEAD = (!FDB || last || 2 * (NW0 + NL0) + 2 > NMFMA) ? 0 : 2;
                constexpr int ROOM = (NMFMA - LEAD) / 2;
                constexpr int NW = (NW0 <= ROOM) ? NW0 : ROOM;
                constexpr int NL = (NW + NL0 <= ROOM) ? NL0 : (ROOM - NW);
                if constexpr (LEAD > 0) __builtin_amdgcn_sched_group_barrier(0x8, LEAD, 0);
                if constexpr (FDB || kg != 0 || !CARRY) __builtin_amdgcn_sched_group_barrier(0x100, NFRAG, 0);
                sched_pairs<0x200, NW>();
                sched_pairs<0x20, NL>();
                if constexpr (NMFMA - LEAD - 2 * (NW + NL) > 0)
                    __builtin_amdgcn_sched_group_barrier(0x8, NMFMA - LEAD - 2 * (NW + NL), 0);
            });
        };
        if constexpr (BUF == 2) {
            int t = 0;
            for (; t + 1 < nk; t += 2) { ktile(t, 0); ktile(t + 1, 1); }
            if (t < nk) ktile(t, 0);
        } else {
            for (int t = 0; t < nk; ++t) ktile(t, t & 1);
        }
        __syncthreads();   // the staging buffers are re-used by the next segment / the epilogue reductions

        // Epilogue.
        if constexpr (epi_uses_lds<Epi>::value) epi.set_lds(smem, wave, WTC, lane);
        epi.begin();
        // Two phases per ROW of MFMA tiles: issue every global load the epilogue needs for the row (prefetch), then
        // compute and store (apply).  (Inputs and outputs of an epilogue may alias as far as the compiler knows, so a
        // fused load-compute-store per element serialises one memory round trip per element.)  The scheduling barrier
        // after each row keeps the compiler from hoisting ALL rows' prefetches to the top: with f64 (16 tiles per
        // wave) that cost 250+ VGPRs and one wave per SIMD.
        if constexpr (!EARLY) prefetch_row(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        static_for<TR>([&](auto IC) {
            constexpr int i = decltype(IC)::value;
            if constexpr (EARLY && i + 1 < TR)
                prefetch_row(std::integral_constant<int, i + 1>{}, std::integral_constant<int, (i + 1) & 1>{});
#pragma unroll
            for (int j = 0; j < TC; ++j)
#pragma unroll
                for (int reg = 0; reg < M::NACC; ++reg)
                    epi.apply(i * MT + reg_row(reg), j * MT, acc[i][j][reg], j, pre[i & 1][j][reg]);
            if constexpr (epi_uses_lds<Epi>::value) epi.row_done(i * MT);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!EARLY && i + 1 < TR)
                prefetch_row(std::integral_constant<int, i + 1>{}, std::integral_constant<int, (i + 1) & 1>{});
        });
        // (the next segment's prologue stages operands into the buffer such an epilogue may still be reading in a slower wave)
        if constexpr (epi_uses_lds<Epi>::value) __syncthreads();
    }
    epi.template finish<MT, TC, WGR, WGC>(reinterpret_cast<double *>(smem), tctx);
}

// ---------------------------------------------------------------------------
// Epilogues: setup(split, tile) once per output tile (descriptors, lane offsets); pre = prefetch(ro, co) loads whatever
// the epilogue needs from memory for the element at (ro, co) relative to the wave tile origin (plus the lane part);
// begin() after the main loop; apply(ro, co, v, jt, pre) receives D and stores.  Element (r, c) lives at base[c + r*ld].
// ---------------------------------------------------------------------------

// C (or split-K slab `split`) = acc.  Tail segments (split < 0, see the kernel's work decomposition) go to a second
// matrix: tail slab p = C2 + p*stride2, element (r, c) at (c - c_off) + (r - r_off)*ld2.
template <typename T> struct EpiStore {
    T *C;
    int64_t ld, slab_stride;
    T *dst;
    T *C2 = nullptr;
    int64_t ld2 = 0, stride2 = 0, r_off = 0, c_off = 0;
    // piece_rows > 0: the main output is BLOCKED along c -- piece g = c / piece_rows is a contiguous (piece_rows x R) column-major
    // matrix at C + g * piece_stride (the reduce-scatter send buffer of the row-sharded W side: piece g = the rows rank g owns);
    // wave tiles never straddle pieces (piece_rows is a multiple of 128)
    int64_t piece_rows = 0, piece_stride = 0;
    rsrc_t rd;
    LaneAddr<T> la;
    struct Pre {};
    static constexpr bool EARLY = false, HEAVY = false;
    __device__ __forceinline__ void setup(int split, const TileCtx &t) {
        int64_t ldc;
        if (split >= 0) {
            dst = C + (int64_t)split * slab_stride; ldc = ld;
            if (piece_rows > 0) {
                const int64_t g = t.cw0 / piece_rows;
                dst += g * piece_stride - g * piece_rows;      // element (r, c) at dst + c + r * ld with ld = piece_rows
            }
        } else { dst = C2 + (int64_t)(-1 - split) * stride2 - (c_off + r_off * ld2); ldc = ld2; }
        rd = tile_rsrc(dst, ldc, t);
        la.init(t, ldc);
    }
    __device__ __forceinline__ void begin() {}
    __device__ __forceinline__ Pre prefetch(int, int) const { return Pre{}; }
    __device__ __forceinline__ void apply(int ro, int co, T v, int /*jt*/, const Pre &) { buf_st(rd, la.lb, la.soff(ro, co), v); }
    template <int MT, int TC, int WGR, int WGC> __device__ __forceinline__ void finish(double *, const TileCtx &) {}
};

// EpiStore whose main output is blocked along c into pieces of piece_rows (as above) that live in DIFFERENT allocations: piece g is the
// (piece_rows x R) column-major matrix at piece[g] -- rank g's receive slot of the peer-to-peer exchange (peer.hpp), mapped into this
// process: the X_g H_g' product stores row block g of the numerator straight into rank g's memory (write-through, system-scope
// stores: they leave this GPU's caches as they are issued), so the reduce-scatter of the row-sharded W side has no send buffer, no
// pack launch and no collective -- only a flag behind the launch.  Tail segments (the Gram riding in the launch) stay local.
constexpr int EPI_MAX_PIECES = 16;
template <typename T> struct EpiStorePeer {
    T *piece[EPI_MAX_PIECES];
    int64_t piece_rows;
    T *C2 = nullptr;
    int64_t ld2 = 0, stride2 = 0, r_off = 0, c_off = 0;
    rsrc_t rd;
    LaneAddr<T> la;
    bool remote;
    // Round 6: the remote stores leave as 16-BYTE write-through stores.  The MFMA result layout gives a lane one element per output row
    // (lanes run along c), so a direct store is 4 bytes per lane -- and a system-scope 4-byte store is a fabric write of its own (the
    // hardware guide prices `dword sc1` stores at ~6x the time per byte of `dwordx4`; over xGMI a 16-byte packet payload is the
    // difference between a fifth and most of a link's rate).  The wave therefore stages every row of MFMA tiles (32 rows x its 64
    // columns) in the block's LDS -- free behind the main loop -- and reads it back row-wise: 16 lanes x 16 bytes = one 256-byte row
    // segment per store instruction.  Float32 with 64-wide wave tiles (the 128 x 128 block tile of the big products); anything else keeps
    // the direct stores.
    // MEASURED ON THE ONE-GPU STAND-IN AND NOT ENABLED (stage16 = false): with every window in local uncached memory the staged form is
    // SLOWER -- X_g H_g' 137.5 -> 147.6 us at the 8-rank shard shape, the sequence 0.330 -> 0.340 ms
    // (profiles/r06_simranks8_p2p_staged_16_byte_peer_stores_rejected_all_events.json): local memory takes 4-byte write-through stores
    // at the rate the epilogue issues them, and the LDS round trip is pure cost.  Whether xGMI changes the sign cannot be measured on
    // this box; the path stays behind the flag for the first run on a node where it can.
    static constexpr bool USES_LDS = true;
    static constexpr int LDS_PITCH = 72;           // floats per staged row: rows 4 apart (the two half-waves of a store) fall into disjoint banks
    T *lds = nullptr;
    bool staged = false;
    bool stage16 = false;                          // NMFX_P2P_STAGE16=1 (development switch, Solver::times_ht)
    int lane_ = 0;
    int64_t ldc_ = 0;
    __device__ __forceinline__ void set_lds(T *smem, int wave, int wtc, int lane) {
        staged = stage16 && sizeof(T) == 4 && wtc == 64;
        lds = smem + wave * (32 * LDS_PITCH);
        lane_ = lane;
    }
    struct Pre {};
    static constexpr bool EARLY = false, HEAVY = false;
    __device__ __forceinline__ void setup(int split, const TileCtx &t) {
        int64_t ldc;
        T *dst;
        if (split >= 0) {
            const int64_t g = t.cw0 / piece_rows;
            dst = piece[g] - g * piece_rows;                    // element (r, c) at dst + c + r * piece_rows
            ldc = piece_rows;
            remote = true;
        } else { dst = C2 + (int64_t)(-1 - split) * stride2 - (c_off + r_off * ld2); ldc = ld2; remote = false; }
        rd = tile_rsrc(dst, ldc, t);
        la.init(t, ldc);
        ldc_ = ldc;
    }
    __device__ __forceinline__ void begin() {}
    __device__ __forceinline__ Pre prefetch(int, int) const { return Pre{}; }
    // behind the apply() calls of the MFMA tile row that starts at wave-tile row `row0`: the staged 32 x 64 block leaves as 16-byte stores
    __device__ __forceinline__ void row_done(int row0) {
        if constexpr (sizeof(T) == 4) {
            if (!(remote && staged)) return;
            typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int idx = lane_ + 64 * q, row = idx >> 4, c4 = (idx & 15) * 4;
                const v4u_t v = *reinterpret_cast<const v4u_t *>(lds + row * LDS_PITCH + c4);
                const uint32_t off = (uint32_t)(((int64_t)(row0 + row) * ldc_ + c4) * (int64_t)sizeof(T));
                __builtin_amdgcn_raw_buffer_store_b128(v, rd, (int)off, 0, 17);
            }
        }
    }
    __device__ __forceinline__ void apply(int ro, int co, T v, int /*jt*/, const Pre &) {
        if (remote) {
            if constexpr (sizeof(T) == 4) {
                if (staged) {
                    // (wave-tile row ro + rl, column co + cl; the lane part is the same for every element: la.lb = (cl + rl * ldc) * 4)
                    const int rl = 4 * (lane_ >> 5), cl = lane_ & 31;
                    lds[((ro & 31) + rl) * LDS_PITCH + co + cl] = v;
                    return;
                }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rd, (int)la.lb, (int)la.soff(ro, co), 17);
            } else {
                typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_t, v), rd, (int)la.lb, (int)la.soff(ro, co), 17);
            }
        } else buf_st(rd, la.lb, la.soff(ro, co), v);
    }
    template <int MT, int TC, int WGR, int WGC> __device__ __forceinline__ void finish(double *, const TileCtx &) {}
};

// Multiplicative update (src/multupd.jl:101-103, 112-114):
//   out = old * ( max(0, num - lambda) / (acc + delta) ),  acc = Gram-form denominator
// num may still be ONE or TWO split-K slabs (summed here in ascending slab order, like reduce_slabs_kernel).
// STATS = 1 additionally accumulates the stop_condition sums of the component that runs along c
// (src/common.jl:100-104: dev = sum (new-old)^2, sum = sum (new+old)^2; term in T, sum in Float64) and writes
// one partial per r-tile:  stat_partial[(tr*ncomp + c)*2 + {0,1}]  -- the layout finalize_partials_kernel reduces.
// NSL = the largest slab count the prefetch handles (2 or 8): slabs 1 .. NSL-1 are loaded unconditionally and selected (see prefetch).
template <typename T, int STATS, int NSL = 2> struct EpiMultUpdate {
    const T *num;
    int nslab;
    int64_t slab_stride;
    const T *old;
    T *out;
    int64_t ld;
    T lambda, delta;
    double *stat_partial;
    int ncomp;
    double dev[STATS ? 8 : 1], sm[STATS ? 8 : 1];
    rsrc_t rnum, rold, rout;
    LaneAddr<T> la;
    int64_t tile_off;
    // STATS == 2 (Float32): the new factor is ALSO written transposed -- element (r, c) at outT[r + c * ldT] -- so that the X*H' product
    // of the W side finds H' contraction-contiguous (Solver::times_ht with the X' image).  The four consecutive rows a lane owns per
    // accumulator group (C/D layout: row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) leave as TWO 8-byte stores.
    // NOT one 16-byte store: `buffer_store_dwordx4 v[a:a+3], voff, s[desc], sN offen` with the scalar offset in an SGPR, followed directly by
    // a VALU write of v[a:a+1] (the registers are dead behind the store and the allocator hands them to the next v_cvt_f64_f32), stored
    // the NEW contents of v[a:a+1] on gfx950 -- sporadically, under load from other queues (found with four ranks sharing the GPU:
    // 64-384 of 327 680 elements per launch carried halves of the statistics' Float64 temporaries).  LLVM inserts the wait states of the
    // ">64-bit store data, then VALU write of the data registers" hazard only when the scalar offset is NOT a register (the rule of
    // earlier chips); with an immediate offset it puts two VALU instructions between and those stores were never wrong.  8-byte
    // stores are outside the hazard class altogether.
    T *outT = nullptr;
    int64_t ldT = 0;
    rsrc_t routT;
    uint32_t lbT, pitchT;
    f32x4 tv;
    __device__ __forceinline__ void setup(int, const TileCtx &t) {
        rnum = tile_rsrc(num, ld, t);
        rold = tile_rsrc(old, ld, t);
        rout = tile_rsrc(out, ld, t);
        tile_off = t.cw0 + t.rw0 * ld;
        la.init(t, ld);
        if constexpr (STATS == 2) {
            routT = __builtin_amdgcn_make_buffer_rsrc((void *)(outT + (t.rw0 + t.cw0 * ldT)), 0, -1, 0x00020000);
            lbT = (uint32_t)(((int64_t)t.rl + (int64_t)t.cl * ldT) * (int64_t)sizeof(T));
            pitchT = (uint32_t)(ldT * (int64_t)sizeof(T));
        }
    }
    __device__ __forceinline__ void begin() {
        if constexpr (STATS != 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { dev[j] = 0.0; sm[j] = 0.0; }
        }
    }
    struct Pre { T nu, ov; };
    static constexpr bool EARLY = true;   // prefetch() does not depend on begin()
    static constexpr bool HEAVY = true;   // 128 x 128 tiles leave one wave per SIMD: always run on half-size tiles
    __device__ __forceinline__ Pre prefetch(int ro, int co) const {
        const uint32_t so = la.soff(ro, co);
        T nu = buf_ld<T>(rnum, la.lb, so);
        // The second slab (the usual case: the numerator product ran 2-way split-K) is loaded UNCONDITIONALLY -- from slab 0 again when
        // there is only one -- and selected afterwards.  As a loop over a run-time slab count every element's load sat in its own
        // basic block behind s_waitcnt vmcnt(0): 16 dependent memory round trips per wave tile in front of the main loop (ISA).
#pragma unroll
        for (int sl = 1; sl < NSL; ++sl) {
            const int64_t s1 = (nslab > sl) ? (int64_t)sl * slab_stride : 0;
            const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(num + (s1 + tile_off)), 0, -1, 0x00020000);
            const T v1 = buf_ld<T>(rs, la.lb, so);
            const T sum = nu + v1;
            nu = (nslab > sl) ? sum : nu;
        }
        // (nslab <= NSL here: the callers combine more slabs by a reduction launch first -- Solver::h_num_nslab / w_num_nslab; NSL = 8 serves
        // the 8-rank shard of the headline problem, whose W'X runs 8-way split-K: the 7 us combine launch and its 2 MB round trip go away)
        return Pre{nu, buf_ld<T>(rold, la.lb, so)};
    }
    __device__ __forceinline__ void apply(int ro, int co, T v, int jt, const Pre &pre) {
        T t = pre.nu - lambda;
        t = (t > (T)0) ? t : ((t != t) ? t : (T)0);   // max(zero(T), t); NaN propagates like Julia's max
        const T ov = pre.ov;
        const T nv = ov * (t / (v + delta));
        buf_st(rout, la.lb, la.soff(ro, co), nv);
        if constexpr (STATS == 2) {
            static_assert(STATS != 2 || sizeof(T) == 4, "transposed copy: Float32 only");
            tv[ro & 1] = nv;
            if ((ro & 1) == 1) {
                typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
                typedef float v2f_t __attribute__((ext_vector_type(2)));
                const v2f_t pr = {tv[0], tv[1]};
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_t, pr), routT, (int)lbT, (int)((uint32_t)(ro - 1) * (uint32_t)sizeof(T) + (uint32_t)co * pitchT), 0);
            }
        }
        if constexpr (STATS != 0) {
            const T d = nv - ov, sp = nv + ov;
            dev[jt] += (double)(T)(d * d);
            sm[jt] += (double)(T)(sp * sp);
        }
    }
    template <int MT, int TC, int WGR, int WGC> __device__ __forceinline__ void finish(double *smem, const TileCtx &t) {
        if constexpr (STATS != 0) {
            static_assert(TC <= 8, "statistics accumulators");
            constexpr int WTC = TC * MT, BCW = WGC * WTC;   // columns per wave / per block
            __syncthreads();                               // smem aliases the GEMM staging buffers
#pragma unroll
            for (int j = 0; j < TC; ++j) {
                double d = dev[j], q = sm[j];
#pragma unroll
                for (int off = 32; off >= MT; off >>= 1) { d += __shfl_down(d, off, 64); q += __shfl_down(q, off, 64); }
                if (t.lane < MT) {
                    const int cl = t.wc * WTC + j * MT + t.lane;
                    smem[(t.wr * BCW + cl) * 2] = d;
                    smem[(t.wr * BCW + cl) * 2 + 1] = q;
                }
            }
            __syncthreads();
            for (int e = t.tid; e < BCW * 2; e += t.nthreads) {
                double s = 0.0;
                for (int w = 0; w < WGR; ++w) s += smem[w * BCW * 2 + e];
                stat_partial[((int64_t)t.tr * ncomp) * 2 + (t.c0) * 2 + e] = s;
            }
        }
    }
};

// The same multiplicative update for the ROW-SHARDED W side of the multi-GPU step: numerator, old factor and
// output each have their own leading dimension -- the numerator is read straight from the reduce-scatter's output (Pc x K piece,
// ld Pc), the old factor from the rank's rows of W (ld P), the new rows are written straight into the rank's chunk of the
// all-gather buffer (ld Pc) -- so no pack / unpack launch surrounds the update.  Same arithmetic, same bits.
template <typename T> struct EpiMultUpdateRows {
    const T *num;
    int64_t ldn;       // numerator AND output (both Pc x K pieces)
    const T *old;
    int64_t ldo;
    T *out;
    T lambda, delta;
    // out2 != nullptr: the new rows ALSO go to a second Pc x K piece with system-scope write-through stores -- the rank's slot in its
    // own exchange window, from which the peers PULL them (solver_impl.hpp: multmse_w_rows_fused_peer; `out` stays the cached copy
    // the rank's own next products read)
    T *out2 = nullptr;
    rsrc_t rnum, rold, rout, rout2;
    LaneAddr<T> ln, lo;
    __device__ __forceinline__ void setup(int, const TileCtx &t) {
        rnum = tile_rsrc(num, ldn, t);
        rout = tile_rsrc(out, ldn, t);
        rold = tile_rsrc(old, ldo, t);
        if (out2 != nullptr) rout2 = tile_rsrc(out2, ldn, t);
        ln.init(t, ldn);
        lo.init(t, ldo);
    }
    __device__ __forceinline__ void begin() {}
    struct Pre { T nu, ov; };
    static constexpr bool EARLY = true, HEAVY = true;
    __device__ __forceinline__ Pre prefetch(int ro, int co) const {
        return Pre{buf_ld<T>(rnum, ln.lb, ln.soff(ro, co)), buf_ld<T>(rold, lo.lb, lo.soff(ro, co))};
    }
    __device__ __forceinline__ void apply(int ro, int co, T v, int /*jt*/, const Pre &pre) {
        T t = pre.nu - lambda;
        t = (t > (T)0) ? t : ((t != t) ? t : (T)0);
        const T nv = pre.ov * (t / (v + delta));
        buf_st(rout, ln.lb, ln.soff(ro, co), nv);
        if (out2 != nullptr) {
            if constexpr (sizeof(T) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, nv), rout2, (int)ln.lb, (int)ln.soff(ro, co), 17);
            else {
                typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_t, nv), rout2, (int)ln.lb, (int)ln.soff(ro, co), 17);
            }
        }
    }
    template <int MT, int TC, int WGR, int WGC> __device__ __forceinline__ void finish(double *, const TileCtx &) {}
};

// out = max(acc, 0)   (projectnn!, src/utils.jl:34-41; NaN passes through)
template <typename T> struct EpiClampStore {
    T *out;
    int64_t ld;
    rsrc_t rout;
    LaneAddr<T> la;
    __device__ __forceinline__ void setup(int, const TileCtx &t) { rout = tile_rsrc(out, ld, t); la.init(t, ld); }
    __device__ __forceinline__ void begin() {}
    struct Pre {};
    static constexpr bool EARLY = false, HEAVY = false;
    __device__ __forceinline__ Pre prefetch(int, int) const { return Pre{}; }
    __device__ __forceinline__ void apply(int ro, int co, T v, int /*jt*/, const Pre &) {
        buf_st(rout, la.lb, la.soff(ro, co), (v < (T)0) ? (T)0 : v);
    }
    template <int MT, int TC, int WGR, int WGC> __device__ __forceinline__ void finish(double *, const TileCtx &) {}
};

// out = max(acc, 0) together with the stop_condition sums of the component that runs along c against `old` (the layout and the
// arithmetic of EpiMultUpdate<T, 1>): ProjectedALS's H solve ends in this epilogue instead of a separate statistics pass.
template <typename T> struct EpiClampStats {
    const T *old;
    T *out;
    int64_t ld;
    double *stat_partial;
    int ncomp;
    double dev[8], sm[8];
    rsrc_t rold, rout;
    LaneAddr<T> la;
    __device__ __forceinline__ void setup(int, const TileCtx &t) { rold = tile_rsrc(old, ld, t); rout = tile_rsrc(out, ld, t); la.init(t, ld); }
    __device__ __forceinline__ void begin() {
#pragma unroll
        for (int j = 0; j < 8; ++j) { dev[j] = 0.0; sm[j] = 0.0; }
    }
    struct Pre { T ov; };
    static constexpr bool EARLY = true, HEAVY = false;
    __device__ __forceinline__ Pre prefetch(int ro, int co) const { return Pre{buf_ld<T>(rold, la.lb, la.soff(ro, co))}; }
    __device__ __forceinline__ void apply(int ro, int co, T v, int jt, const Pre &pre) {
        const T nv = (v < (T)0) ? (T)0 : v;
        buf_st(rout, la.lb, la.soff(ro, co), nv);
        const T d = nv - pre.ov, sp = nv + pre.ov;
        dev[jt] += (double)(T)(d * d);
        sm[jt] += (double)(T)(sp * sp);
    }
    template <int MT, int TC, int WGR, int WGC> __device__ __forceinline__ void finish(double *smem, const TileCtx &t) {
        static_assert(TC <= 8, "statistics accumulators");
        constexpr int WTC = TC * MT, BCW = WGC * WTC;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TC; ++j) {
            double d = dev[j], q = sm[j];
#pragma unroll
            for (int off = 32; off >= MT; off >>= 1) { d += __shfl_down(d, off, 64); q += __shfl_down(q, off, 64); }
            if (t.lane < MT) {
                const int cl = t.wc * WTC + j * MT + t.lane;
                smem[(t.wr * BCW + cl) * 2] = d;
                smem[(t.wr * BCW + cl) * 2 + 1] = q;
            }
        }
        __syncthreads();
        for (int e = t.tid; e < BCW * 2; e += t.nthreads) {
            double s = 0.0;
            for (int w = 0; w < WGR; ++w) s += smem[w * BCW * 2 + e];
            stat_partial[((int64_t)t.tr * ncomp) * 2 + (t.c0) * 2 + e] = s;
        }
    }
};

// out = acc - sub   (projected-gradient G = Gram*Z - B, src/alspgrad.jl:124-127, 280-283)
template <typename T> struct EpiSubStore {
    const T *sub;
    T *out;
    int64_t ld;
    rsrc_t rsub, rout;
    LaneAddr<T> la;
    __device__ __forceinline__ void setup(int, const TileCtx &t) { rsub = tile_rsrc(sub, ld, t); rout = tile_rsrc(out, ld, t); la.init(t, ld); }
    __device__ __forceinline__ void begin() {}
    struct Pre { T s; };
    static constexpr bool EARLY = true, HEAVY = false;
    __device__ __forceinline__ Pre prefetch(int ro, int co) const { return Pre{buf_ld<T>(rsub, la.lb, la.soff(ro, co))}; }
    __device__ __forceinline__ void apply(int ro, int co, T v, int /*jt*/, const Pre &pre) { buf_st(rout, la.lb, la.soff(ro, co), v - pre.s); }
    template <int MT, int TC, int WGR, int WGC> __device__ __forceinline__ void finish(double *, const TileCtx &) {}
};

// Q = X ./ (acc + delta)   (src/multupd.jl:172-174, 184-186), acc = (W*H) tile kept in registers
// x / d for the ratio pass, d = (WH)_ij + delta > 0.  FAST (Float32): v_rcp_f32's 1-ulp reciprocal and ONE residual correction,
//   r = rcp(d);  q0 = x r;  q = fma(fma(-d, q0, x), r, q0)
// -- 4 vector instructions instead of the 12 of the correctly rounded IEEE sequence (v_div_scale x 2, v_rcp, five fma, v_div_fmas,
// v_div_fixup), which sit in the shadow of the matrix-core instructions of the W*H product and cost it 8 %.
// The residual is formed exactly by the fused multiply-add, so q is the correctly rounded quotient except on near-ties (then one ulp
// off), and exact whenever x / d is representable; no scaling for operands near the ends of the exponent range (d >= delta =
// sqrt(eps) here; a quotient beyond 2^126 or a d beyond 2^126 comes out as inf / 0 / NaN where IEEE division still returns a finite value).
// nmfx_opts-independent switch: NMFX_DIV_IEEE=1 in the environment keeps the IEEE sequence.
__device__ __forceinline__ float ratio_div_fast(float x, float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    const float q0 = x * r;
    return __builtin_fmaf(__builtin_fmaf(-d, q0, x), r, q0);
}
__device__ __forceinline__ double ratio_div_fast(double x, double d) { return x / d; }

template <typename T, int FAST = 0> struct EpiRatio {
    const T *X;
    T *Q;
    int64_t ld;
    T delta;
    rsrc_t rx, rq;
    LaneAddr<T> la;
    __device__ __forceinline__ void setup(int, const TileCtx &t) { rx = tile_rsrc(X, ld, t); rq = tile_rsrc(Q, ld, t); la.init(t, ld); }
    __device__ __forceinline__ void begin() {}
    struct Pre { T x; };
    static constexpr bool EARLY = true, HEAVY = false;
    __device__ __forceinline__ Pre prefetch(int ro, int co) const { return Pre{buf_ld<T>(rx, la.lb, la.soff(ro, co))}; }
    __device__ __forceinline__ void apply(int ro, int co, T v, int /*jt*/, const Pre &pre) {
        if constexpr (FAST != 0) buf_st(rq, la.lb, la.soff(ro, co), ratio_div_fast(pre.x, v + delta));
        else buf_st(rq, la.lb, la.soff(ro, co), pre.x / (v + delta));
    }
    template <int MT, int TC, int WGR, int WGC> __device__ __forceinline__ void finish(double *, const TileCtx &) {}
};

// block-level sum of per-thread doubles in a fixed order -> *dst
__device__ __forceinline__ void block_sum_store(double v, double *smem, int tid, int nthreads, double *dst) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();   // smem may alias a staging buffer: every wave is past its last LDS read here
    if ((tid & 63) == 0) smem[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < nthreads / 64; ++w) s += smem[w];
        *dst = s;
    }
}

__device__ __forceinline__ float nmfx_log(float x) { return logf(x); }
__device__ __forceinline__ double nmfx_log(double x) { return log(x); }

// Objective partials without materialising WH (evaluate_objv):
//   KL == 0: sum (x - acc)^2           term in T, sum in Float64 (StatsBase.sqL2dist; src/multupd.jl:81)
//   KL == 1: sum x>0 ? x*log(x/acc) - x + acc : acc          (StatsBase.gkldiv;  src/multupd.jl:148)
template <typename T, int KL> struct EpiObjective {
    const T *X;
    int64_t ld;
    double *partial;   // one per block
    double sum;
    rsrc_t rx;
    LaneAddr<T> la;
    __device__ __forceinline__ void setup(int, const TileCtx &t) { rx = tile_rsrc(X, ld, t); la.init(t, ld); }
    __device__ __forceinline__ void begin() { sum = 0.0; }
    struct Pre { T x; };
    static constexpr bool EARLY = true, HEAVY = false;
    __device__ __forceinline__ Pre prefetch(int ro, int co) const { return Pre{buf_ld<T>(rx, la.lb, la.soff(ro, co))}; }
    __device__ __forceinline__ void apply(int, int, T v, int /*jt*/, const Pre &pre) {
        const T x = pre.x;
        T t;
        if constexpr (KL == 0) {
            const T d = x - v;
            t = d * d;
        } else {
            if (x > (T)0) t = x * nmfx_log(x / v) - x + v;
            else t = v;
        }
        sum += (double)t;
    }
    template <int MT, int TC, int WGR, int WGC> __device__ __forceinline__ void finish(double *smem, const TileCtx &t) {
        block_sum_store(sum, smem, t.tid, t.nthreads, partial + t.bid);
    }
};

}  // namespace nmfx
