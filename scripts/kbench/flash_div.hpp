// flash_div.hpp -- EXPERIMENT, not part of libnmfx.so (DESIGN.md section 3.2 "K5"): the divergence updates of
// MultUpdate(obj = :div) without ever materialising Q = X ./ (WH + delta) (src/multupd.jl:172-175 and :184-187; SURVEY.md
// section 7 "K5").  Correct (flash_bench.hip checks it against a brute-force fp64 kernel) but, at one wave per SIMD, no faster
// than the shipped two-launch path: 2.41 ms (W side) / 2.29 ms (H side) at the C3 shape against 1.25 + 1.00 ms for
// gemm_WH_ratio + gemm_XHt/WtX -- 73-76 % of the fp32 MFMA peak.  Kept with its harness as the starting point for a
// hand-scheduled version; what was measured on the way is listed in DESIGN.md.
//
//
//     H side:  WtQ(a, j) = sum_i W(i, a) * X(i, j) / ((W H)(i, j) + delta)
//     W side:  QHt(i, a) = sum_j X(i, j) / ((W H)(i, j) + delta) * H(a, j)
//
// Both are ONE kernel, written once in operand-neutral form.  With a RESIDENT operand R(x, b) (64 values of x per block,
// all k components b), a STREAMED operand S(y, b) (tiles of 64 values of y) and the data Xm(x, y), x contiguous:
//
//     Out(x, a) = sum_y  [ Xm(x, y) / ( sum_b R(x, b) S(y, b) + delta ) ] * S(y, a)
//
//     W side: x = i, y = j, R = W (component index strided), S = H (component index contiguous), Xm = X,  Out = QHt
//     H side: x = j, y = i, R = H (contiguous),              S = W (strided),                    Xm = X', Out = WtQ
//
// (the H side reads the transposed copy X' that the solver keeps for this algorithm: it replaces the p x n buffer Q the
// round-1 path wrote and re-read twice per iteration, so the footprint is unchanged and 4 p n sizeof(T) bytes of HBM traffic
// per iteration disappear).
//
// Per block and y-tile two chained MFMA GEMMs of equal size run out of LDS, 128 KiB for K = 256: the whole R panel stays
// resident for the life of the block, the S tile is loaded ONCE and feeds both products:
//     GEMM1   D1(y, x) = sum_b S(y, b) R(x, b)               (v_mfma_f32_32x32x2_f32, lanes along x)
//     ratio   q(x, y)  = Xm(x, y) / (D1 + delta)             in the accumulator registers (x along lanes: coalesced X loads)
//     GEMM2   D2(a, x) += sum_y S(y, a) q(x, y)              q is fed to the MFMA as the B operand STRAIGHT from the accumulator
//                                                            registers of GEMM1: in the 32x32 C/D layout register s of lane
//                                                            (n, h) holds row 4h + (s&3) + 8(s>>2), and taking exactly those rows
//                                                            as the two k-slots of step s makes acc[s] the B operand of step s.
// No k-loop staging, no Q tile in LDS, two barriers per y-tile; the next S tile travels global -> registers under the MFMAs.
// 4 waves = (2 x-tiles) x (2 y-tiles); D2 partials of the two y-halves are added through LDS once, at the very end.
// f32, K in {64, 128, 256}; other shapes keep the materialised-Q path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_mfma.hpp"

#ifdef FLASH_DBG_FASTDIV
#define FLASH_DIV(a, b) ((a) * __builtin_amdgcn_rcpf(b))
#else
#define FLASH_DIV(a, b) ((a) / (b))
#endif

namespace nmfx {
namespace flash {

struct Args {
    const float *R;    // resident operand
    int64_t ldr;
    const float *S;    // streamed operand
    int64_t lds;
    const float *X;    // Xm(x, y) at X[y*ldx + x]
    int64_t ldx;
    float *Out;        // slab s at Out + s*slab_stride; element (x, a) at a*ldo + x  (A_CONTIG = 0)  or  x*ldo + a  (A_CONTIG = 1)
    int64_t ldo, slab_stride;
    int xblocks;       // 64-wide x blocks
    int ytiles;        // 64-wide y tiles handled by ONE block (per split)
    float delta;
    const int *done;
};

// 16-byte chunk c of row r of a [64 rows][K floats] image lives at chunk position c ^ rot(r): 16 consecutive rows reading the
// same chunk, and 16 consecutive chunks of one row, both fall on 16 distinct 16-byte bank groups (256 B = all 64 banks).
__device__ __forceinline__ int rot16(int r) { return r & 15; }
template <int K> __device__ __forceinline__ int chunk_off(int r, int c) { return (r * (K / 4) + (c ^ rot16(r))) * 4; }   // in floats

// global -> registers of one [64 x K] operand tile (rows row0.., all K components), 16 bytes per load
template <int K, bool KSTRIDED_SRC> struct OperandTile {
    static constexpr int PER = K / 16;     // 16-byte loads per thread (256 threads)
    __device__ static __forceinline__ void load(f32x4 (&v)[PER], const float *base, int64_t ld, int64_t row0, int tid) {
        if constexpr (!KSTRIDED_SRC) {
            // element (r, b) at base[(row0 + r)*ld + b]: chunk q -> row q / (K/4), chunk q % (K/4); a wave reads whole rows
#pragma unroll
            for (int m = 0; m < PER; ++m) {
                const int q = tid + 256 * m, r = q / (K / 4), c = q % (K / 4);
                v[m] = *reinterpret_cast<const f32x4 *>(base + (row0 + r) * ld + 4 * c);
            }
        } else {
            // element (r, b) at base[b*ld + row0 + r]: 4 x 4 micro-tiles (4 rows x 4 components); lane -> (b4 = lane & 3, r4 = lane >> 2)
            // so that the 16 lanes of a b128 LDS write pass hit 16 distinct bank groups; v[4*m + e] = rows 4*r4..+3 of component 4*b4 + e
            const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
            for (int m = 0; m < PER / 4; ++m) {
                const int b4 = (lane & 3) + 4 * wave + 16 * m, r4 = lane >> 2;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * m + e] = *reinterpret_cast<const f32x4 *>(base + (int64_t)(4 * b4 + e) * ld + row0 + 4 * r4);
            }
        }
    }
    // the m-th of the PER loads of load() alone (the main loop issues them one at a time between MFMA groups)
    __device__ static __forceinline__ f32x4 load_one(int m, const float *base, int64_t ld, int64_t row0, int tid) {
        if constexpr (!KSTRIDED_SRC) {
            const int q = tid + 256 * m, r = q / (K / 4), c = q % (K / 4);
            return *reinterpret_cast<const f32x4 *>(base + (row0 + r) * ld + 4 * c);
        } else {
            const int lane = tid & 63, wave = tid >> 6;
            const int b4 = (lane & 3) + 4 * wave + 16 * (m / 4), r4 = lane >> 2;
            return *reinterpret_cast<const f32x4 *>(base + (int64_t)(4 * b4 + (m % 4)) * ld + row0 + 4 * r4);
        }
    }
    __device__ static __forceinline__ void store(const f32x4 (&v)[PER], float *img, int tid) {
        if constexpr (!KSTRIDED_SRC) {
#pragma unroll
            for (int m = 0; m < PER; ++m) {
                const int q = tid + 256 * m, r = q / (K / 4), c = q % (K / 4);
                *reinterpret_cast<f32x4 *>(img + chunk_off<K>(r, c)) = v[m];
            }
        } else {
            const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
            for (int m = 0; m < PER / 4; ++m) {
                const int b4 = (lane & 3) + 4 * wave + 16 * m, r4 = lane >> 2;
#pragma unroll
                for (int er = 0; er < 4; ++er) {
                    f32x4 t;
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = v[4 * m + e][er];
                    *reinterpret_cast<f32x4 *>(img + chunk_off<K>(4 * r4 + er, b4)) = t;
                }
            }
        }
    }
};

template <int K, bool R_KSTRIDED, bool S_KSTRIDED, bool A_CONTIG>
__global__ __launch_bounds__(256, 1) void flash_div_kernel(Args g) {
    static_assert(K == 64 || K == 128 || K == 256, "component count (padded) must be 64, 128 or 256");
    if (g.done != nullptr && *reinterpret_cast<const volatile int *>(g.done) != 0) return;
    constexpr int NV = (K >= 128) ? 4 : 2;       // a-tiles fed by one LDS read of the S row (a = GA*u + NV*n + v)
    constexpr int GA = 32 * NV;                  // components per a-group
    constexpr int NU = K / GA;                   // a-groups
    using vecv_t = float __attribute__((ext_vector_type(NV)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Rimg = smem, *Simg = smem + 64 * K;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xt = wave & 1, yt = wave >> 1;
    const int n = lane & 31, h = lane >> 5;
    const int xb = blockIdx.x % g.xblocks, split = blockIdx.x / g.xblocks;
    const int64_t x0 = (int64_t)xb * 64;
    const int64_t ybeg = (int64_t)split * g.ytiles * 64;

    using LoadR = OperandTile<K, R_KSTRIDED>;
    using LoadS = OperandTile<K, S_KSTRIDED>;
    f32x4 sreg[LoadS::PER];
    {
        f32x4 rreg[LoadR::PER];
        LoadR::load(rreg, g.R, g.ldr, x0, tid);
        LoadS::load(sreg, g.S, g.lds, ybeg, tid);
        LoadR::store(rreg, Rimg, tid);
        LoadS::store(sreg, Simg, tid);
    }
    __syncthreads();

    f32x16 acc2[NU][NV];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[u][v][r] = 0.f;

    // the lane's column of Xm and the 16 rows it owns in the wave's 32 x 32 tile: row(s) = 4h + (s&3) + 8(s>>2)
    const float *xcol = g.X + (ybeg + 32 * yt + 4 * h) * g.ldx + x0 + 32 * xt + n;
    const int rS = 32 * yt + n, rR = 32 * xt + n;           // this lane's rows of the S / R images in GEMM1

    constexpr int NG = K / 8;                               // k-groups of GEMM1 (4 MFMAs each)
    constexpr int NLD = 16 + LoadS::PER;                    // global loads per y-tile: 16 Xm values + the next S tile
    constexpr int LPG = (NLD + NG - 1) / NG;                // ... issued LPG per k-group of GEMM1
    // The block runs ONE wave per SIMD (128 KiB of LDS), and a wave issues in order: an MFMA that finds the matrix pipe busy
    // stalls everything behind it.  Non-MFMA work therefore only overlaps the ~60-cycle shadow of the MFMA issued right
    // before it, and the loop is written as small fenced regions { a few loads / LDS reads / VALU, then 2-4 MFMAs }
    // (sched_barrier(0) keeps the compiler from gathering the loads at the top, where nothing would cover them).
    for (int t = 0; t < g.ytiles; ++t) {
        float xv[16];
        const float *xc = xcol;
        xcol += 64 * g.ldx;
        // next S tile (the last iteration re-loads its own tile: no branch in the loop body)
        const int tn = (t + 1 < g.ytiles) ? t + 1 : t;
        const int64_t srow = ybeg + (int64_t)tn * 64;

        // ---- GEMM1: D1(y, x) = sum_b S(y, b) R(x, b); lane's k-slot h and step q <-> b = 8g + 4h + q
        f32x16 acc1;      // (four rotating accumulators, summed at the end, measured 6 % SLOWER than this single chain)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
        f32x4 fa[2], fb[2];
        fa[0] = *reinterpret_cast<const f32x4 *>(Simg + chunk_off<K>(rS, h));
        fb[0] = *reinterpret_cast<const f32x4 *>(Rimg + chunk_off<K>(rR, h));
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            if (gq + 1 < NG) {
                fa[(gq + 1) & 1] = *reinterpret_cast<const f32x4 *>(Simg + chunk_off<K>(rS, 2 * (gq + 1) + h));
                fb[(gq + 1) & 1] = *reinterpret_cast<const f32x4 *>(Rimg + chunk_off<K>(rR, 2 * (gq + 1) + h));
            }
#pragma unroll
            for (int l = 0; l < LPG; ++l) {
                const int id = gq * LPG + l;     // this tile's global loads, a few per k-group
                if (id < 16) xv[id] = xc[(int64_t)((id & 3) + 8 * (id >> 2)) * g.ldx];
                else if (id < NLD) sreg[id - 16] = LoadS::load_one(id - 16, g.S, g.lds, srow, tid);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[gq & 1][q], fb[gq & 1][q], acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- ratio in the accumulators (src/multupd.jl:172-174, :184-186) and
        // ---- GEMM2: D2(a, x) += sum_y S(y, a) q(x, y); step s contracts rows y = 4h + (s&3) + 8(s>>2) of the wave's y-tile
        vecv_t sv[2][NU];
        float qv[2];
        qv[0] = FLASH_DIV(xv[0], (acc1[0] + g.delta));
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int a0 = GA * u + NV * n;      // a = GA*u + NV*n + v, v = 0..NV-1: NV consecutive floats of chunk a0 / 4
            sv[0][u] = *reinterpret_cast<const vecv_t *>(Simg + chunk_off<K>(32 * yt + 4 * h, a0 >> 2) + (a0 & 3));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int ry = 32 * yt + 4 * h + ((s + 1) & 3) + 8 * ((s + 1) >> 2);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (s + 1 < 16) {
                    const int a0 = GA * u + NV * n;
                    sv[(s + 1) & 1][u] = *reinterpret_cast<const vecv_t *>(Simg + chunk_off<K>(ry, a0 >> 2) + (a0 & 3));
                    if (u == NU - 1) qv[(s + 1) & 1] = FLASH_DIV(xv[s + 1], (acc1[s + 1] + g.delta));
                }
#ifndef FLASH_DBG_NO_GEMM2
#pragma unroll
                for (int v = 0; v < NV; ++v) acc2[u][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[s & 1][u][v], qv[s & 1], acc2[u][v], 0, 0, 0);
#else
                acc2[0][0][s] += sv[s & 1][u][0] * qv[s & 1];
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#ifdef FLASH_DBG_NO_BARRIER
        LoadS::store(sreg, Simg, tid);
#elif !defined(FLASH_DBG_NO_STORE)
        __syncthreads();                       // every wave is done reading this S tile
        LoadS::store(sreg, Simg, tid);
        __syncthreads();
#else
        acc2[0][0][0] += sreg[0][0] + sreg[LoadS::PER - 1][3];
#endif
    }

    // ---- epilogue: partials of the two y-halves -> LDS -> sum -> global.  LDS image: A_CONTIG ? [x][K + 4] : [a][64 + 4]
    // (every wave is past its last read of the operand images: the loop ends with a barrier)
    constexpr int LDX = 64 + 4, LDA = K + 4;
    float *part = smem + (size_t)yt * (A_CONTIG ? 64 * LDA : K * LDX);
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * h;      // D row
                const int a = GA * u + NV * m + v, x = 32 * xt + n;
                if constexpr (A_CONTIG) part[x * LDA + a] = acc2[u][v][r];
                else part[a * LDX + x] = acc2[u][v][r];
            }
    __syncthreads();
    float *out = g.Out + (int64_t)split * g.slab_stride;
    const float *p0 = smem, *p1 = smem + (A_CONTIG ? 64 * LDA : K * LDX);
    if constexpr (A_CONTIG) {
        // Out[(x0 + x)*ldo + a]: 4 consecutive a per thread
        for (int e = tid; e < 64 * K / 4; e += 256) {
            const int x = e / (K / 4), a = 4 * (e % (K / 4));
            const f32x4 s0 = *reinterpret_cast<const f32x4 *>(p0 + x * LDA + a), s1 = *reinterpret_cast<const f32x4 *>(p1 + x * LDA + a);
            *reinterpret_cast<f32x4 *>(out + (x0 + x) * g.ldo + a) = s0 + s1;
        }
    } else {
        // Out[a*ldo + x0 + x]: 4 consecutive x per thread
        for (int e = tid; e < 64 * K / 4; e += 256) {
            const int a = e / 16, x = 4 * (e % 16);
            const f32x4 s0 = *reinterpret_cast<const f32x4 *>(p0 + a * LDX + x), s1 = *reinterpret_cast<const f32x4 *>(p1 + a * LDX + x);
            *reinterpret_cast<f32x4 *>(out + (int64_t)a * g.ldo + x0 + x) = s0 + s1;
        }
    }
}

template <int K> constexpr size_t lds_bytes() {
    // operand images 2 * 64 * K floats; epilogue 2 partial images of max(64*(K+4), K*(64+4)) floats
    constexpr size_t a = (size_t)2 * 64 * K, b = (size_t)2 * (K * 68 > 64 * (K + 4) ? K * 68 : 64 * (K + 4));
    return (a > b ? a : b) * sizeof(float);
}

// Xt(j, i) = X(i, j): 64 x 64 tiles through LDS; X is P x N (ld P), Xt is N x P (ld N); P, N multiples of 64
__global__ __launch_bounds__(256) void transpose_kernel(float *Xt, const float *X, int64_t P, int64_t N) {
    __shared__ float tile[64][65];
    const int64_t i0 = (int64_t)blockIdx.x * 64, j0 = (int64_t)blockIdx.y * 64;
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int i = e & 63, j = e >> 6;
        tile[j][i] = X[(j0 + j) * P + i0 + i];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int j = e & 63, i = e >> 6;
        Xt[(i0 + i) * N + j0 + j] = tile[j][i];
    }
}

}  // namespace flash
}  // namespace nmfx
