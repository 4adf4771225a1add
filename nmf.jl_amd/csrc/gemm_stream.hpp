// gemm_stream.hpp -- the short-K, large-output product W*H with a fused epilogue, as ONE persistent software pipeline.
//
//   D(r = j, c = i) = sum_a H(a, j) W(i, a)        (never materialised; src/multupd.jl:104,115,172-174,184-186; :81,148)
//
// with Q = X ./ (WH + delta) (the ratio pass of MultUpdate(:div)) or the objective terms reduced in the epilogue.  The
// contraction is only K = 128 ... 1024 long while the output is p x n: per 128 x 128 output tile the block-per-tile kernel
// (gemm_mfma.hpp) runs 4-32 k-tiles of MFMAs and then an epilogue that touches 64-128 KiB of HBM -- a memory phase during
// which the wave issues no MFMA, and since both blocks of a CU start together they reach it together (measured at the C3
// shape: plain store 1043 us, objective 1119 us, ratio 1254 us against an MFMA floor of 874 us).
//
// Here a block is PERSISTENT (2 per CU) and walks a sequence of output tiles as one continuous stream of k-tiles: the
// operand loads of the next tile's first k-tiles are issued under the last k-tiles of the current one, the finished
// accumulators are handed to a second register set, and that tile's epilogue -- X loads, the division / objective term,
// the Q stores -- is issued a few elements per k-group UNDER the MFMAs of the next tile's first four k-tiles.  No wave
// ever has an MFMA-free phase besides the one barrier per k-tile, and the epilogue's HBM traffic is spread evenly over
// the launch instead of arriving in bursts.
//
// f32, A = H (K-contiguous), B = W (K-strided), K a multiple of 128 (the solver's padding rule for k > 64).
// LDS images, fragment reads and the MFMA order are those of gemm_mfma.hpp, so every accumulator holds the same bits.
#pragma once
#include "gemm_mfma.hpp"

namespace nmfx {

struct StreamArgs {
    const float *A;   // H: K x N, ld = K      rows of the operand <-> r (columns of X)
    const float *B;   // W: P x K, ld = P      rows of the operand <-> c (rows of X, contiguous in the output)
    int64_t lda, ldb;
    int tiles_r, tiles_c;   // 128 x 128 output tiles
    int nkt;                // K / 32, a multiple of 4
    int group;              // 8: 8 x 8 super-tile rasterisation (both tile counts multiples of 8), else 1
    const int *done;
};

// Epilogues of the stream kernel.  Per-tile state (the buffer descriptors at the wave tile origin) is a separate small object,
// because two tiles are alive at any time: the one being accumulated and the one whose epilogue is running.  The work on one
// element is cut into three STAGES of a few VALU instructions each; the kernel issues one stage behind each MFMA, so a wave's
// epilogue arithmetic runs in the 64-cycle shadow of its own MFMAs instead of forming MFMA-free runs (with two in-order waves
// per SIMD sharing the matrix pipe, every cycle in which BOTH are inside such a run is a lost pipe cycle: measured +7 % on the
// ratio pass with the 12-instruction division issued as one run per element).
template <typename T, int FAST = 0> struct SEpiRatio {   // Q = X ./ (acc + delta); FAST: ratio_div_fast's arithmetic (gemm_mfma.hpp), cut into the same three stages
    const T *X;
    T *Q;
    int64_t ld;
    T delta;
    LaneAddr<T> la;
    struct Tile { rsrc_t rx, rq; };
    struct Pre { T x; };
    struct St { T d, ds, ns, r, f1, mul, f2; bool vcc; };
    __device__ __forceinline__ void init(const TileCtx &t) { la.init(t, ld); }
    __device__ __forceinline__ Tile tile(int64_t rw0, int64_t cw0) const {
        Tile t;
        t.rx = __builtin_amdgcn_make_buffer_rsrc((void *)(X + (cw0 + rw0 * ld)), 0, -1, 0x00020000);
        t.rq = __builtin_amdgcn_make_buffer_rsrc((void *)(Q + (cw0 + rw0 * ld)), 0, -1, 0x00020000);
        return t;
    }
    __device__ __forceinline__ Pre prefetch(const Tile &t, int ro, int co) const { return Pre{buf_ld<T>(t.rx, la.lb, la.soff(ro, co))}; }
    // x / d, correctly rounded: the instruction sequence the compiler expands an IEEE Float32 division into (v_div_scale x 2,
    // v_rcp, the Newton / residual fma chain, v_div_fmas, v_div_fixup), written out so that it can be issued in three pieces.
    // Bit-identical to `x / d` (checked element by element against the block-per-tile kernel: scripts/kbench/stream_bench.hip).
    template <int S> __device__ __forceinline__ void stage(const Tile &t, int ro, int co, T v, const Pre &p, St &st) {
        static_assert(sizeof(T) == 4, "f32");
        if constexpr (FAST != 0) {
            if constexpr (S == 0) {
                st.d = v + delta;
                st.r = __builtin_amdgcn_rcpf(st.d);
            } else if constexpr (S == 1) {
                st.mul = p.x * st.r;
                st.f2 = __builtin_fmaf(-st.d, st.mul, p.x);
            } else {
                buf_st(t.rq, la.lb, la.soff(ro, co), __builtin_fmaf(st.f2, st.r, st.mul));
            }
        } else if constexpr (S == 0) {
            st.d = v + delta;
            bool unused;
            st.ds = __builtin_amdgcn_div_scalef(p.x, st.d, false, &unused);
            st.ns = __builtin_amdgcn_div_scalef(p.x, st.d, true, &st.vcc);
            st.r = __builtin_amdgcn_rcpf(st.ds);
        } else if constexpr (S == 1) {
            const T f0 = __builtin_fmaf(-st.ds, st.r, 1.0f);
            st.f1 = __builtin_fmaf(f0, st.r, st.r);
            st.mul = st.ns * st.f1;
            st.f2 = __builtin_fmaf(-st.ds, st.mul, st.ns);
        } else {
            const T f3 = __builtin_fmaf(st.f2, st.f1, st.mul);
            const T f4 = __builtin_fmaf(-st.ds, f3, st.ns);
            const T q = __builtin_amdgcn_div_fmasf(f4, st.f1, f3, st.vcc);
            buf_st(t.rq, la.lb, la.soff(ro, co), __builtin_amdgcn_div_fixupf(q, st.d, p.x));
        }
    }
    __device__ __forceinline__ void finish(double *, int, int, int) {}
};

template <typename T, int KL> struct SEpiObjective {   // sum (x - acc)^2  or the KL term; term in T, sum in Float64
    const T *X;
    int64_t ld;
    double *partial;   // one per block
    double sum;
    LaneAddr<T> la;
    struct Tile { rsrc_t rx; };
    struct Pre { T x; };
    struct St { T t; };
    __device__ __forceinline__ void init(const TileCtx &t) { la.init(t, ld); sum = 0.0; }
    __device__ __forceinline__ Tile tile(int64_t rw0, int64_t cw0) const {
        Tile t;
        t.rx = __builtin_amdgcn_make_buffer_rsrc((void *)(X + (cw0 + rw0 * ld)), 0, -1, 0x00020000);
        return t;
    }
    __device__ __forceinline__ Pre prefetch(const Tile &t, int ro, int co) const { return Pre{buf_ld<T>(t.rx, la.lb, la.soff(ro, co))}; }
    template <int S> __device__ __forceinline__ void stage(const Tile &, int, int, T v, const Pre &p, St &st) {
        const T x = p.x;
        if constexpr (KL == 0) {
            if constexpr (S == 0) {
                const T d = x - v;
                st.t = d * d;
            } else if constexpr (S == 1) {
                sum += (double)st.t;
                // keep the term where it is written: nothing else orders this pure arithmetic, and sunk to the end of the tile it
                // would hold every X value and accumulator of the tile in registers (seen: 256 registers + scratch)
                asm volatile("" : "+v"(sum));
            }
        } else {
            if constexpr (S == 0) {
                st.t = x / v;
                asm volatile("" : "+v"(st.t));
            } else if constexpr (S == 1) {
                st.t = nmfx_log(st.t);
                asm volatile("" : "+v"(st.t));
            } else {
                T t;
                if (x > (T)0) t = x * st.t - x + v;
                else t = v;
                sum += (double)t;
                asm volatile("" : "+v"(sum));
            }
        }
    }
    __device__ __forceinline__ void finish(double *smem, int tid, int nthreads, int bid) { block_sum_store(sum, smem, tid, nthreads, partial + bid); }
};

template <typename Epi, int LEADCH = 2>
__global__ __launch_bounds__(256, 2) void gemm_wh_stream_kernel(StreamArgs g, Epi epi) {
    using T = float;
    using M = Mfma<T>;
    constexpr int BR = 128, BC = 128, BK = 32, NT = 256, WGC = 2, WTR = 64, WTC = 64, TR = 2, TC = 2, MT = 32, NG = 4, UB = 4;
    using LoadA = TileLoader<T, KCONTIG, BR, NT>;
    using LoadB = TileLoader<T, KSTRIDED, BC, NT>;
    static_assert(LoadA::PER_THREAD == 4 && LoadB::PER_THREAD == 4, "staging registers");
    using vec_t = typename M::vec_t;
    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));

    if (g.done != nullptr && *reinterpret_cast<const volatile int *>(g.done) != 0) return;
    __shared__ __attribute__((aligned(16))) T smem[2 * (BR + BC) * BK];
    constexpr int STAGE = (BR + BC) * BK;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WGC, wc = wave % WGC;

    // ---- tile sequence.  Blocks are dealt round-robin to the 8 XCDs: block b of the launch sits on XCD b % 8, slot b / 8.
    // Step s of the walk gives the 64 blocks of XCD x the 8 x 8 super-tile  s*8 + x  (one tile each), so an XCD's L2 holds
    // the 8 + 8 operand panels of one super-tile at a time (2 MiB at K = 256) and every panel is fetched once per XCD and step.
    const int nblk = gridDim.x;
    int lid = blockIdx.x;
    if ((nblk & 7) == 0) lid = (lid & 7) * (nblk >> 3) + (lid >> 3);
    const int tiles = g.tiles_r * g.tiles_c;
    auto tile_origin = [&](int tix, int64_t &r0, int64_t &c0) {
        int tr, tc;
        if (g.group > 1) {
            const int G = g.group, per = G * G;
            const int st = tix / per, in = tix % per;
            const int sr = st / (g.tiles_c / G), sc = st % (g.tiles_c / G);
            tr = sr * G + in / G;
            tc = sc * G + in % G;
        } else { tc = tix % g.tiles_c; tr = tix / g.tiles_c; }
        r0 = (int64_t)__builtin_amdgcn_readfirstlane(tr) * BR;
        c0 = (int64_t)__builtin_amdgcn_readfirstlane(tc) * BC;
    };
    if (lid >= tiles) return;

    // ---- operand loads: buffer loads with a per-k-tile descriptor (scalar) and loop-invariant 32-bit lane offsets
    uint32_t voffA[4], voffB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = tid + NT * i, row = s / 8, cpos = s % 8, c = cpos ^ swz8(row);
        voffA[i] = (uint32_t)(((int64_t)row * g.lda + c * 4) * 4);
    }
    {
        const int kq = tid / 32, r4 = tid % 32;   // micro-tile (k-quad, row-quad) of this thread
#pragma unroll
        for (int ek = 0; ek < 4; ++ek) voffB[ek] = (uint32_t)((((int64_t)(kq * 4 + ek)) * g.ldb + r4 * 4) * 4);
    }
    vec_t ra[4], rb[4];
    auto load_tiles = [&](const T *Ak, const T *Bk) {   // Ak = A + r0*lda + k0, Bk = B + c0 + k0*ldb
        const rsrc_t da = __builtin_amdgcn_make_buffer_rsrc((void *)Ak, 0, -1, 0x00020000);
        const rsrc_t db = __builtin_amdgcn_make_buffer_rsrc((void *)Bk, 0, -1, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(da, (int)voffA[i], 0, 0));
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(db, (int)voffB[i], 0, 0));
    };
    auto load_A = [&](const T *Ak) {
        const rsrc_t da = __builtin_amdgcn_make_buffer_rsrc((void *)Ak, 0, -1, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(da, (int)voffA[i], 0, 0));
    };
    auto load_B = [&](const T *Bk) {
        const rsrc_t db = __builtin_amdgcn_make_buffer_rsrc((void *)Bk, 0, -1, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(db, (int)voffB[i], 0, 0));
    };

    TileCtx tctx{0, 0, wr, wc, lane, tid, NT, (int)blockIdx.x, 0, 0, 4 * (lane >> 5), lane % MT, 0, 0};
    epi.init(tctx);
    auto reg_row = [](int reg) { return (reg & 3) + 8 * (reg >> 2); };

    // ---- state of the walk
    int tix = lid;
    int64_t r0, c0, nr0, nc0;
    tile_origin(tix, r0, c0);
    int ntix = tix + nblk;
    bool has_next = ntix < tiles;
    if (has_next) tile_origin(ntix, nr0, nc0); else { nr0 = r0; nc0 = c0; }
    const T *Ab = g.A + r0 * g.lda, *Bb = g.B + c0;
    const T *An = g.A + nr0 * g.lda, *Bn = g.B + nc0;
    typename Epi::Tile tcur = epi.tile(r0 + wr * WTR, c0 + wc * WTC), tprev = tcur;

    typename M::acc_t acc[TR][TC], eacc[TR][TC];
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j)
#pragma unroll
            for (int r = 0; r < M::NACC; ++r) { acc[i][j][r] = (T)0; eacc[i][j][r] = (T)0; }

    // ---- pipeline prologue: k-tile 0 -> LDS stage 0, k-tile 1 -> registers
    load_tiles(Ab, Bb);
    LoadA::store(ra, smem, tid);
    LoadB::store(rb, smem + BR * BK, tid);
    load_tiles(Ab + BK, Bb + (int64_t)BK * g.ldb);
    __syncthreads();
    T af[2][TR][M::VEC], bf[2][TC][M::VEC];
#pragma unroll
    for (int i = 0; i < TR; ++i) read_frag<T, KCONTIG, BR, NT>(af[0][i], smem, wr * WTR + i * MT, 0, lane);
#pragma unroll
    for (int j = 0; j < TC; ++j) read_frag<T, KSTRIDED, BC, NT>(bf[0][j], smem + BR * BK, wc * WTC + j * MT, 0, lane);

    // epilogue elements: e = (i*TC + j)*16 + reg, cut into NPARTS parts: part p is prefetched in k-group (chunk) p of a body and
    // applied in chunk p + LEADCH
    constexpr int NEL = TR * TC * M::NACC;   // 64
    constexpr int NCH = UB * NG, NPARTS = NCH - LEADCH, PBASE = NEL / NPARTS, PREM = NEL % NPARTS;
    typename Epi::Pre pre[NEL];

    // single staging instructions (the k-group schedule below places them one by one)
    auto load_A1 = [&](const rsrc_t &da, int i) { ra[i] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(da, (int)voffA[i], 0, 0)); };
    auto load_B1 = [&](const rsrc_t &db, int i) { rb[i] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(db, (int)voffB[i], 0, 0)); };
    auto store_A1 = [&](T *lds, int i) { *reinterpret_cast<vec_t *>(lds + (tid + NT * i) * 4) = ra[i]; };
    const int kqB = tid / 32, r4B = tid % 32;
    auto store_B1 = [&](T *lds, int er) {   // row er of the thread's 4 x 4 micro-tile, transposed in registers (TileLoader::store)
        vec_t t;
#pragma unroll
        for (int ek = 0; ek < 4; ++ek) t[ek] = rb[ek][er];
        *reinterpret_cast<vec_t *>(lds + (kqB * BC + swzrow(r4B * 4 + er)) * 4) = t;
    };

    // One BODY = UB = 4 consecutive k-tiles (16 k-groups) of the current tile.  kb = index of the body inside the tile.
    // EPI: the epilogue of the previous tile (eacc, tprev) runs under it.  ZERO: the first MFMA of every accumulator starts
    // from C = 0 (first body of a tile).
    //
    // Issue order of a k-group = program order: the 16 MFMAs are written out one by one, each followed by its share of the
    // k-group's other work and a scheduling fence (sched_barrier(0)), so the compiler cannot gather that work into MFMA-free runs:
    //   behind MFMA 0            : the next k-group's four fragment reads
    //   behind MFMA 1 ... 15     : stage (m-1) % 3 of epilogue element (m-1) / 3 of this k-group (<= 5 elements per k-group)
    //   behind MFMA 2, 5, 8, 11  : one staging instruction each (LDS write of the tile in registers / global load of the tile after it)
    //   behind MFMA 3, 6, ..., 15: prefetch of one epilogue element (X load), LEADCH k-groups ahead of its use
    auto body = [&](auto EPI_C, auto ZERO_C, int kb, bool lastb) {
        constexpr bool EPI = decltype(EPI_C)::value, ZERO = decltype(ZERO_C)::value;
        // operand addresses of the k-tiles this body loads: k-tile kb*4 + kt + 2
        const T *A_same = Ab + (int64_t)(kb * UB + 2) * BK, *B_same = Bb + (int64_t)(kb * UB + 2) * BK * g.ldb;
        // next body's first two k-tiles: of this tile, or (last body) of the NEXT tile
        const T *A_nb = lastb ? An : Ab + (int64_t)(kb * UB + 4) * BK;
        const T *B_nb = lastb ? Bn : Bb + (int64_t)(kb * UB + 4) * BK * g.ldb;
        static_for<UB>([&](auto KTC) {
            constexpr int kt = decltype(KTC)::value;
            constexpr int cur = kt & 1;
            const T *a_s = smem + cur * STAGE, *b_s = a_s + BR * BK;
            T *a_n = smem + (cur ^ 1) * STAGE, *b_n = a_n + BR * BK;
            const T *Ald = (kt < 2) ? A_same + (int64_t)kt * BK : A_nb + (int64_t)(kt - 2) * BK;
            const T *Bld = (kt < 2) ? B_same + (int64_t)kt * BK * g.ldb : B_nb + (int64_t)(kt - 2) * BK * g.ldb;
            const rsrc_t dA = __builtin_amdgcn_make_buffer_rsrc((void *)Ald, 0, -1, 0x00020000);
            const rsrc_t dB = __builtin_amdgcn_make_buffer_rsrc((void *)Bld, 0, -1, 0x00020000);
            static_for<NG>([&](auto KGC) {
                constexpr int kg = decltype(KGC)::value;
                constexpr int ch = kt * NG + kg;
                constexpr bool last = (kg == NG - 1);
                constexpr int fc = kg & 1, fn = fc ^ 1;
                constexpr int NPF = (EPI && ch < NPARTS) ? (ch < PREM ? PBASE + 1 : PBASE) : 0;
                constexpr int pf0 = (ch < PREM) ? (PBASE + 1) * ch : (PBASE + 1) * PREM + PBASE * (ch - PREM);
                constexpr int pa = ch - LEADCH;
                constexpr int NAP = (EPI && ch >= LEADCH) ? (pa < PREM ? PBASE + 1 : PBASE) : 0;
                constexpr int ap0 = (pa < PREM) ? (PBASE + 1) * pa : (PBASE + 1) * PREM + PBASE * (pa - PREM);
                static_assert(PBASE + 1 <= 5, "five epilogue slots per k-group");
                typename Epi::St est[PBASE + 1];
                if constexpr (last) __syncthreads();   // every LDS read of this k-tile and every LDS write of the next one is issued
                static_for<16>([&](auto MC) {
                    constexpr int m = decltype(MC)::value;
                    constexpr int q = m / (TR * TC), i = (m % (TR * TC)) / TC, j = m % TC;
                    if (ZERO && kt == 0 && kg == 0 && q == 0) {
                        typename M::acc_t z;
#pragma unroll
                        for (int r = 0; r < M::NACC; ++r) z[r] = (T)0;
                        acc[i][j] = M::mma(af[fc][i][q], bf[fc][j][q], z);
                    } else {
                        acc[i][j] = M::mma(af[fc][i][q], bf[fc][j][q], acc[i][j]);
                    }
                    if constexpr (m == 0) {   // the next k-group's fragments (the next k-tile's first group behind the barrier)
                        const T *fa = last ? a_n : a_s, *fb = last ? b_n : b_s;
                        constexpr int gn = last ? 0 : kg + 1;
#pragma unroll
                        for (int ii = 0; ii < TR; ++ii) read_frag<T, KCONTIG, BR, NT>(af[fn][ii], fa, wr * WTR + ii * MT, gn, lane);
#pragma unroll
                        for (int jj = 0; jj < TC; ++jj) read_frag<T, KSTRIDED, BC, NT>(bf[fn][jj], fb, wc * WTC + jj * MT, gn, lane);
                    }
                    if constexpr (m >= 1) {
                        constexpr int x = (m - 1) / 3, sg = (m - 1) % 3;
                        if constexpr (x < NAP) {
                            constexpr int e = ap0 + x;
                            constexpr int t = e / M::NACC, reg = e % M::NACC, ei = t / TC, ej = t % TC;
                            epi.template stage<sg>(tprev, ei * MT + ((reg & 3) + 8 * (reg >> 2)), ej * MT, eacc[ei][ej][reg], pre[e], est[x]);
                        }
                    }
                    if constexpr (m >= 2 && m <= 11 && m % 3 == 2) {   // 2, 5, 8, 11
                        constexpr int si = (m - 2) / 3;
                        if constexpr (kg == 0) store_A1(a_n, si);
                        if constexpr (kg == 1) store_B1(b_n, si);
                        if constexpr (kg == 2) load_A1(dA, si);
                        if constexpr (kg == 3) load_B1(dB, si);
                    }
                    if constexpr (m >= 3 && m % 3 == 0) {   // 3, 6, 9, 12, 15
                        constexpr int x = m / 3 - 1;
                        if constexpr (x < NPF) {
                            constexpr int e = pf0 + x;
                            constexpr int t = e / M::NACC, reg = e % M::NACC, ei = t / TC, ej = t % TC;
                            pre[e] = epi.prefetch(tprev, ei * MT + ((reg & 3) + 8 * (reg >> 2)), ej * MT);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        });
    };
    (void)reg_row;

    const int nb = g.nkt / UB;
    using TrueC = std::integral_constant<bool, true>;
    using FalseC = std::integral_constant<bool, false>;
    // first tile: nothing to drain yet
    for (int kb = 0; kb < nb; ++kb) body(FalseC{}, FalseC{}, kb, kb == nb - 1);
    for (;;) {
        // hand the finished tile over to the epilogue set
#pragma unroll
        for (int i = 0; i < TR; ++i)
#pragma unroll
            for (int j = 0; j < TC; ++j) eacc[i][j] = acc[i][j];
        tprev = tcur;
        if (!has_next) break;
        // advance
        tix = ntix; r0 = nr0; c0 = nc0; Ab = An; Bb = Bn;
        tcur = epi.tile(r0 + wr * WTR, c0 + wc * WTC);
        ntix = tix + nblk;
        has_next = ntix < tiles;
        if (has_next) { tile_origin(ntix, nr0, nc0); An = g.A + nr0 * g.lda; Bn = g.B + nc0; }
        body(TrueC{}, TrueC{}, 0, nb == 1);
        for (int kb = 1; kb < nb; ++kb) body(FalseC{}, FalseC{}, kb, kb == nb - 1);
    }
    // drain: the last tile's epilogue, two phases per row of MFMA tiles like the block-per-tile kernel
    __syncthreads();
    static_for<TR>([&](auto IC) {
        constexpr int i = decltype(IC)::value;
#pragma unroll
        for (int j = 0; j < TC; ++j)
#pragma unroll
            for (int reg = 0; reg < M::NACC; ++reg) pre[(i * TC + j) * M::NACC + reg] = epi.prefetch(tprev, i * MT + ((reg & 3) + 8 * (reg >> 2)), j * MT);
#pragma unroll
        for (int j = 0; j < TC; ++j)
#pragma unroll
            for (int reg = 0; reg < M::NACC; ++reg) {
                typename Epi::St st;
                const int ro = i * MT + ((reg & 3) + 8 * (reg >> 2)), co = j * MT;
                const typename Epi::Pre &pp = pre[(i * TC + j) * M::NACC + reg];
                epi.template stage<0>(tprev, ro, co, eacc[i][j][reg], pp, st);
                epi.template stage<1>(tprev, ro, co, eacc[i][j][reg], pp, st);
                epi.template stage<2>(tprev, ro, co, eacc[i][j][reg], pp, st);
            }
        __builtin_amdgcn_sched_barrier(0);
    });
    epi.finish(reinterpret_cast<double *>(smem), tid, NT, (int)blockIdx.x);
}

}  // namespace nmfx
