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

#include <utility>

#include "gemm_mfma.hpp"
#include "kernels.hpp"

namespace nmfx {
#ifdef NMFX_POTRF_TIMING_LIB
// (instrumented build only: scripts/r06_potrf_coresident_stamps.sh -- phase stamps of the last potrf_reg_kernel launch)
static __device__ long long nmfx_potrf_dbg[2048];
#endif

__device__ __forceinline__ float nmfx_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double nmfx_sqrt(double x) { return sqrt(x); }

// wave-uniform broadcast of lane `src` (compile-time constant): v_readlane, no LDS crossbar round trip
__device__ __forceinline__ float lane_bcast(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ __forceinline__ double lane_bcast(double v, int src) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// ---- helpers shared by the factorisation kernels (static loops, pins, the pivot, the 32-step elimination of one wave) ----
template <typename F, int... I> __device__ __forceinline__ void strip_static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void strip_static_for(F &&f) { strip_static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{}); }
// empty asm statements that pin values to a place in the instruction stream (as functions: an asm operand inside a generic lambda cannot
// name a variable of the enclosing function)
template <typename A> __device__ __forceinline__ void pin_v(A &a) { asm volatile("" : "+v"(a)); }
template <typename A, typename B> __device__ __forceinline__ void pin_v(A &a, B &b) { asm volatile("" : "+v"(a), "+v"(b)); }
template <typename A, typename B, typename C> __device__ __forceinline__ void pin_v(A &a, B &b, C &c) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c)); }
template <int FROM, int N, typename T> __device__ __forceinline__ void pin_from(T (&c)[N]) {
    if constexpr (FROM < N) {
        asm volatile("" : "+v"(c[FROM]));
        pin_from<FROM + 1>(c);
    }
}
__device__ __forceinline__ float nmfx_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double nmfx_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
// s = sqrt(d), r = 1 / s for a pivot.  Float32: the hardware's 1-ulp estimates (v_sqrt_f32, v_rcp_f32) and one Newton step each with
// the residuals formed exactly by fused multiply-adds -- a chain of 6 dependent instructions instead of the 28 of the correctly rounded
// sqrtf + division sequences, which were a third of the elimination's 32 dependent steps.  Exact where the arithmetic is (a perfect
// square gives residual 0 or a correction that rounds onto the root), within an ulp of the correctly rounded values otherwise (near
// ties only); pivots in the Float32 denormal range come out non-finite and are reported as not positive definite.
__device__ __forceinline__ void pivot_sqrt_rcp(float d, float &s, float &r) {
    const float s0 = __builtin_amdgcn_sqrtf(d);
    const float r0 = __builtin_amdgcn_rcpf(s0);
    const float e = __builtin_fmaf(-s0, s0, d);
    s = __builtin_fmaf(e, 0.5f * r0, s0);
    const float e2 = __builtin_fmaf(-s, r0, 1.0f);
    r = __builtin_fmaf(r0, e2, r0);
}
__device__ __forceinline__ void pivot_sqrt_rcp(double d, double &s, double &r) {
    s = sqrt(d);
    r = 1.0 / s;
}

// The 32 right-looking elimination steps on one wave: lanes 0-31 hold the columns of a 32 x 32 diagonal block (rows 0 .. lane meaningful),
// lanes 32-63 the columns of a panel block to its right (or a copy of the diagonal block's), 32 rows each in col[].  On return the diagonal
// lanes hold U_bb (upper triangle), the panel lanes inv(U_bb') times their columns, myrinv of lane j < 32 is 1 / U_bb(j, j), bad is set
// on a non-positive pivot.  Shared by potrf_reg_kernel and potrf_upper_kernel.
template <typename T> __device__ __forceinline__ void chol_eliminate32(T (&col)[32], int lane, T &myrinv, bool &bad) {
    constexpr int NB = 32;
    // Software-pipelined over the steps: the pivot of step j + 1 (broadcast, square root, reciprocal: a chain of ten dependent
    // instructions with two transcendental latencies) is started as soon as column j + 1 has its update from step j, and runs
    // under the 30 - j independent updates of the other columns.
    T dj, rj;
    {
        const T d0 = lane_bcast(col[0], 0);
        bad = !(d0 > (T)0);
        pivot_sqrt_rcp(d0, dj, rj);
    }
    strip_static_for<NB>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        int lj = lane;
        pin_v(lj);     // (keeps the 32 masks lane == j out of the scalar registers: hoisted out of the block loop they spill)
        const bool diag = lj == j;
        if (diag) myrinv = rj;
        col[j] = diag ? dj : col[j] * rj;
        if constexpr (j + 1 < NB) {
            col[j + 1] = nmfx_fma(-lane_bcast(col[j], j + 1), col[j], col[j + 1]);
            const T d = lane_bcast(col[j + 1], j + 1);
            bad = bad || !(d > (T)0);
            pivot_sqrt_rcp(d, dj, rj);
        }
        // (the broadcasts eight at a time into scalar registers BEFORE the updates that use them: a scalar written by v_readlane is
        // not available to the next vector instruction for several cycles, and broadcast / update pairs issued back to back ran at
        // ~16 cycles per pair)
        strip_static_for<(NB - j - 2 > 0 ? (NB - j - 2 + 7) / 8 : 0)>([&](auto gc_) {
            constexpr int r0 = j + 2 + 8 * decltype(gc_)::value;
            T m[8];
            strip_static_for<8>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                if constexpr (r0 + u < NB) m[u] = lane_bcast(col[j], r0 + u); else m[u] = (T)0;
            });
            asm volatile("" : "+s"(m[0]), "+s"(m[1]), "+s"(m[2]), "+s"(m[3]), "+s"(m[4]), "+s"(m[5]), "+s"(m[6]), "+s"(m[7]));
            strip_static_for<8>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                if constexpr (r0 + u < NB) col[r0 + u] = nmfx_fma(-m[u], col[j], col[r0 + u]);
            });
        });
        // RIGHT-looking as written: the updates of a step are independent of each other.  Left alone, instruction selection sinks
        // every update to its use (left-looking: column r takes its r updates as one dependent chain right before its own pivot,
        // with the multipliers of all earlier steps parked in -- and spilled from -- scalar registers); the empty asm statements
        // pin each updated value, and the next pivot, to their step
        pin_from<j + 2>(col);
        pin_v(dj, rj);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// A: k x k leading block of a column-major matrix with leading dimension ld.  On a non-positive pivot sets
// ctrl->status = posdef_status and ctrl->done = 1 (PosDefException of potrf!, src/utils.jl:68,78).
// Dynamic LDS: 32*32 (diagonal block) + 32*kps (row panel; kps = max(32, kp - 32), kp = k rounded up to 32: the panel right of
// the first diagonal block is the widest) elements of T (32 KiB for k = 256 in f32).
// Per 32-column panel:  (1) wave 0 factors the diagonal block in registers (lane = column, v_readlane
// broadcasts);  (2) the 32 x m row panel is staged through LDS with coalesced loads and solved one column per
// thread;  (3) the trailing update A22 -= R'R runs on the matrix cores (MFMA 32x32x2 f32 / 16x16x4 f64) with
// both operands read from the LDS panel -- the same [k][row] image the GEMM template calls KSTRIDED.
// GPANEL: the row panel lives in a global scratch buffer (`gpanel`, 32 x kps elements, L2-resident) instead of LDS -- the fallback for
// k beyond what one workgroup's LDS holds (k > 1248 in Float32, 608 in Float64): slower (every panel access is a global round trip,
// ordered inside the workgroup by the barriers), but the reference's potrf! has no size limit either.
template <typename T, bool GPANEL = false>
__global__ __launch_bounds__(1024) void potrf_upper_kernel(T *A, int64_t ld, int k, Ctrl *ctrl, int posdef_status, T *gpanel = nullptr) {
    if (ctrl != nullptr && ctrl->done) return;
    // ProjectedALS runs this workgroup on a CU it shares with a block of the big product (projals_impl.hpp): its waves are the
    // YOUNGER ones on their SIMDs and lose every issue arbitration against the GEMM's waves (4x slower than alone).  The
    // factorisation is a latency chain that needs few issue slots, the GEMM a throughput kernel that has plenty: raise the wave
    // priority (priority outranks age, MI355X_MICROARCH.md "Two waves per SIMD").
    __builtin_amdgcn_s_setprio(3);
    using M = Mfma<T>;
    constexpr int NB = 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char chol_smem[];
    const int kp0 = (k + 31) / 32 * 32;
    const int kp = (kp0 > 64) ? kp0 - 32 : 32;          // row stride of the panel image (>= the widest panel, m <= k - 32)
    T *D11 = reinterpret_cast<T *>(chol_smem);          // D11[l*NB + c] = A(jb+l, jb+c) of the current diagonal block, identity-padded
    T *Rp = GPANEL ? gpanel : D11 + NB * NB;            // Rp[l*kp + c]  = U(jb+l, jb+nb+c), zero for c >= m
    // (the flag lives in element (l = 1, c = 0) of the diagonal block's image -- below the diagonal: staged as 0 every step, never used by
    // the elimination -- so that the kernel needs no static LDS on top of the dynamic 160 KiB of the largest panel)
    int *failp = reinterpret_cast<int *>(D11 + NB);
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nwaves = nt >> 6;
    if (tid == 0) *failp = 0;
    __syncthreads();
#ifdef NMFX_POTRF_TIMING
#ifndef NMFX_POTRF_TIMING_LIB
    extern __device__ long long nmfx_potrf_dbg[];
#endif
#define PT(i) if (tid == 0) nmfx_potrf_dbg[(jb / NB) * 8 + (i)] = (long long)__builtin_readcyclecounter();
#else
#define PT(i)
#endif
    for (int jb = 0; jb < k; jb += NB) {
        const int nb = (k - jb < NB) ? (k - jb) : NB;
        const int m = k - jb - nb;
        const int mp = (m + 31) / 32 * 32;
        PT(0)
        // stage the row panel (nb x m) into LDS, zero-padded to (NB x mp): thread <-> (l = tid%32, c = tid/32 + ...), and the diagonal
        // block, a partial one padded with the identity (the 32 elimination steps are branch-free straight-line code; entries below the
        // diagonal are never used)
        for (int e0 = tid; e0 < NB * mp; e0 += 8 * nt) {      // (eight trips' loads in flight, unconditional on clamped addresses)
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * nt, l = e % NB, c = e / NB;
                const bool in = e < NB * mp && l < nb && c < m;
                v[u] = A[in ? (jb + l) + (int64_t)(jb + nb + c) * ld : 0];
                v[u] = in ? v[u] : (T)0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * nt;
                if (e < NB * mp) Rp[(e % NB) * kp + e / NB] = v[u];
            }
        }
        for (int e = tid; e < NB * NB; e += nt) {
            const int l = e % NB, c = e / NB;
            const bool in = l < nb && c < nb && l <= c;
            const T v = A[in ? (jb + l) + (int64_t)(jb + c) * ld : 0];
            D11[l * NB + c] = in ? v : ((l == c) ? (T)1 : (T)0);
        }
        __syncthreads();
        PT(1)
        // Round 6: diagonal block and row panel in ONE instruction stream per wave (chol_eliminate32, as potrf_reg_kernel): wave w takes the
        // diagonal block in lanes 0-31 and 32 panel columns in lanes 32-63 -- every wave repeats the diagonal block's 32 elimination steps
        // (same instructions, same bits) and gets the triangular solve of its panel columns out of the same stream.  Before, one wave
        // factored the diagonal block (17-27 k cycles), THEN the panel was solved one column per thread (12.6 k cycles).
        {
            const int ngroups = (mp > 0) ? mp / 32 : 1;
            for (int pg = wave; pg < ngroups; pg += nwaves) {
                const bool panel = (m > 0) && lane >= NB;
                const int c = 32 * pg + lane - NB;
                T col[NB];
#pragma unroll
                for (int l = 0; l < NB; ++l) col[l] = panel ? Rp[l * kp + (panel ? c : 0)] : D11[l * NB + (lane & 31)];
                T myrinv = (T)0;
                bool bad = false;
                chol_eliminate32(col, lane, myrinv, bad);
                if (panel) {
#pragma unroll
                    for (int l = 0; l < NB; ++l) Rp[l * kp + c] = col[l];
                }
                if (pg == 0 && lane < NB) {
#pragma unroll
                    for (int r = 0; r < NB; ++r)
                        if (lane < nb && r <= lane) A[(jb + r) + (int64_t)(jb + lane) * ld] = col[r];
                    if (bad && lane == 0) *failp = 1;
                }
            }
        }
        __syncthreads();
        PT(2)
        if (*failp) break;
        PT(3)
        // write the solved panel back (coalesced along l) ...
        for (int e = tid; e < NB * m; e += nt) {
            const int l = e % NB, c = e / NB;
            if (l < nb) A[(jb + l) + (int64_t)(jb + nb + c) * ld] = Rp[l * kp + c];
        }
        // ... and update the trailing matrix on the matrix cores: for tiles (ti <= tj) of size MT x MT
        //   A22(ri, cj) -= sum_l R'(l, ri) R'(l, cj).   MFMA lanes run along the matrix ROW index (contiguous in memory).
        {
            constexpr int MT = M::MT, KS = M::KS;
            const int nt_t = mp / MT;
            const int ntiles = nt_t * (nt_t + 1) / 2;
            const int ks = lane / MT, li = lane % MT;
            auto acc_row = [&](int reg) { return (sizeof(T) == 4) ? ((reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) : ((lane >> 4) + 4 * reg); };
            // (round 6) TWO tiles of a wave in flight -- the read-modify-write of a tile is a global round trip the wave has nothing
            // else to do under -- and the loads of the old values UNCONDITIONAL on clamped addresses (under the per-lane triangle test
            // every load was its own basic block and was waited for before the next one was issued); the test selects afterwards
            for (int tile0 = wave; tile0 < ntiles; tile0 += 2 * nwaves) {
                int tis[2], tjs[2];
                bool live[2];
                typename M::acc_t acc[2];
                T oldv[2][M::NACC];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int tile = tile0 + h * nwaves;
                    live[h] = tile < ntiles;
                    // unrank (ti <= tj) from the linear index over the upper triangle, row by row of tj
                    int tj = 0, rem = live[h] ? tile : 0;
                    while (rem > tj) { rem -= tj + 1; ++tj; }
                    tis[h] = rem; tjs[h] = tj;
                    T *base = A + (jb + nb + tis[h] * MT + li) + (int64_t)(jb + nb) * ld;
#pragma unroll
                    for (int reg = 0; reg < M::NACC; ++reg) oldv[h][reg] = base[(int64_t)(tjs[h] * MT + acc_row(reg)) * ld];   // (inside the padded K x K buffer)
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int r = 0; r < M::NACC; ++r) acc[h][r] = (T)0;
#pragma unroll
                    for (int kk = 0; kk < NB / KS; ++kk) {
                        const T a = Rp[(kk * KS + ks) * kp + tjs[h] * MT + li];   // D rows <-> matrix column index cj
                        const T b = Rp[(kk * KS + ks) * kp + tis[h] * MT + li];   // D cols (lanes) <-> matrix row index ri
                        acc[h] = M::mma(a, b, acc[h]);
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int ri = tis[h] * MT + li;
                    T *base = A + (jb + nb + ri) + (int64_t)(jb + nb) * ld;
#pragma unroll
                    for (int reg = 0; reg < M::NACC; ++reg) {
                        const int cj = tjs[h] * MT + acc_row(reg);
                        if (live[h] && ri <= cj && cj < m) base[(int64_t)cj * ld] = oldv[h][reg] - acc[h][reg];
                    }
                }
            }
        }
        PT(4)
        __syncthreads();
        PT(5)
    }
    if (*failp && tid == 0 && ctrl != nullptr) {
        ctrl->status = posdef_status;
        ctrl->done = 1;
    }
}

// ---------------------------------------------------------------------------------------------
// potrf! with the trailing matrix RESIDENT IN REGISTERS (round 6).  potrf_upper_kernel above keeps A in global memory: every block step
// re-reads its diagonal block and row panel from L2 and sends the trailing update through a global read-modify-write -- 20 us per step
// at k = 256, of which the one-wave factorisation of the 32 x 32 diagonal block is 7-11 us and the one-column-per-thread panel solve
// 5 us (profiles/r04_potrf_bench_standalone.log).  Here one workgroup of 8 waves owns the whole upper triangle as 32 x 32 blocks in MFMA
// accumulator layout (block t = bj (bj + 1) / 2 + bi lives in slot t / 8 of wave t % 8: 36 blocks = 5 slots x 16 registers at k = 256
// in Float32), global memory is read once and written once, and a block step is
//   A  the owners of block row b write their blocks to LDS as column images (E[block][column][row]);
//   B  ELIMINATION: wave w < NBLK - 1 - b takes the diagonal block in lanes 0-31 and panel block (b, b + 1 + w) in lanes 32-63, lane =
//      column, the 32 rows of the column in registers, and runs the 32 right-looking steps  u_j = a_j / sqrt(a_jj),  a_r -= u_jr u_j  on
//      all 64 lanes at once: the multipliers u_jr are v_readlane broadcasts from the diagonal lanes.  Every wave repeats the factorisation
//      of the diagonal block (same instructions, same bits) and gets the triangular solve of ITS 32 panel columns out of the very same
//      instruction stream -- the panel solve costs nothing beyond the diagonal block's dependency chain, instead of following it.
//      The scaling multiplies by 1 / sqrt(a_jj) (one correctly rounded division per step, as OpenBLAS' potf2 / trsm kernels do) and the
//      updates are fused multiply-adds.  The solved panel goes to LDS in the MFMA operand image Rp[l][column] and to global memory.
//      An idle wave inverts the PREVIOUS diagonal block meanwhile (Dinv != nullptr: what trtri_diag_kernel computes, one launch less).
//   C  trailing update: every wave subtracts R'R from the blocks it owns, 16 v_mfma_f32_32x32x2 per block, operands from Rp.
// Two workgroup barriers per block step, no global round trip inside the loop.  adddiag! (src/utils.jl:15-24) is fused into the load.
// 512 threads: the workgroup fits the half of a CU that ProjectedALS' short-grid products leave free (projals_impl.hpp).
// NBLK = K / 32 at compile time (the accumulator slots are indexed statically): Float32 up to 8 (k <= 256), Float64 up to 4.
// ---------------------------------------------------------------------------------------------
// (lds_barrier(), kernels.hpp: the factorisation's stores of finished parts of U are never read back inside the kernel, so its barriers must
// not wait for them)

template <typename T, int NBLK> struct PotrfReg {
    static constexpr int NW = 8, NB = 32, NT = NBLK * (NBLK + 1) / 2, NS = (NT + NW - 1) / NW;
    static constexpr int LDE = NB + 16 / (int)sizeof(T);            // column stride of a column image (16-byte aligned rows of 32 + pad)
    static constexpr int KP = (NBLK > 1 ? NBLK - 1 : 1) * NB + 1;   // row stride of the panel image (odd: read along a column without bank conflicts too)
    static constexpr size_t lds_bytes() { return ((size_t)(NBLK + 1) * NB * LDE + (size_t)NB * KP + NBLK * NB) * sizeof(T) + 16; }
};

// Upper triangle (diagonal included) of a 32 x 32 block from its LDS column image Ub[c * LDE + r] to global memory G[r + c * ld], columns
// c < ncols only.  Store INSTRUCTIONS are what costs here (one wave, ~130 cycles each whatever their width): 16-byte stores with
// 32 / V lanes per column cover every V-row group that lies wholly above the diagonal (32 / (2 V) instructions), one or two scalar store
// instructions the V x V triangles on the diagonal -- 6 instructions in Float32 instead of one per pair of columns (16).
template <typename T, int LDE> __device__ __forceinline__ void store_upper32(T *G, int64_t ld, const T *Ub, int ncols, int lane) {
    constexpr int V = 16 / (int)sizeof(T), LPC = 32 / V, CPI = 64 / LPC, NI = 32 / CPI;
    typedef T vec_t __attribute__((ext_vector_type(V)));
    const int q = lane % LPC, cc = lane / LPC;
    vec_t v[NI];
#pragma unroll
    for (int it = 0; it < NI; ++it) v[it] = *reinterpret_cast<const vec_t *>(Ub + (CPI * it + cc) * LDE + V * q);
    constexpr int TRI = V * (V + 1) / 2, NP = (32 / V * TRI + 63) / 64;
    T d[NP];
    int dr[NP], dc[NP];
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
        const int idx = lane + 64 * ps, sb = idx / TRI, e = idx % TRI;      // element e of the triangle of diagonal sub-block sb
        const int c = (e >= 1) + (e >= 3) + (e >= 6);      // (V <= 4: e < 10)
        dr[ps] = (sb < 32 / V) ? V * sb + e - c * (c + 1) / 2 : 32;
        dc[ps] = V * sb + c;
        d[ps] = (sb < 32 / V) ? Ub[dc[ps] * LDE + dr[ps]] : (T)0;
    }
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int c = CPI * it + cc;
        if (V * q + V - 1 < V * (c / V) && c < ncols) *reinterpret_cast<vec_t *>(G + V * q + (int64_t)c * ld) = v[it];      // the whole group above sub-block c / V
    }
#pragma unroll
    for (int ps = 0; ps < NP; ++ps)
        if (dr[ps] < 32 && dc[ps] < ncols) G[dr[ps] + (int64_t)dc[ps] * ld] = d[ps];
}

#ifdef NMFX_POTRF_TIMING
#ifndef NMFX_POTRF_TIMING_LIB
extern __device__ long long nmfx_potrf_dbg[];
#endif
#endif
template <typename T, int NBLK>
__global__ __launch_bounds__(512) void potrf_reg_kernel(T *A, int64_t ld, int k, T lambda, T *Dinv, Ctrl *ctrl, int posdef_status) {
    if (ctrl != nullptr && ctrl->done) return;
    __builtin_amdgcn_s_setprio(3);       // co-resident with a block of the big product: see potrf_upper_kernel
    using M = Mfma<T>;
    using P = PotrfReg<T, NBLK>;
    using vec_t = typename M::vec_t;
    constexpr int NW = P::NW, NB = P::NB, NT = P::NT, NS = P::NS, LDE = P::LDE, KP = P::KP;
    constexpr int MT = M::MT, KS = M::KS, SUB = NB / MT, V = M::VEC;
    extern __shared__ __attribute__((aligned(16))) unsigned char chol_smem[];
    // NBLK + 1 column images of 32 x 32 blocks: step b reads block row b from slots 0 .. NBLK - 1 - b (slot bj - b: E[c * LDE + r] =
    // A(32 b + r, 32 bj + c)); the factored diagonal block b' is kept in slot NBLK - b' (never a slot a later step's block row uses) for
    // the copy to global memory and for its inverse, which overwrites it there
    T *E = reinterpret_cast<T *>(chol_smem);
    T *Rp = E + (size_t)(NBLK + 1) * NB * LDE;        // Rp[l * KP + c] = U(32 b + l, 32 (b + 1) + c)
    T *rinvs = Rp + (size_t)NB * KP;                  // rinvs[b * NB + j] = 1 / U(32 b + j, 32 b + j)
    int *failp = reinterpret_cast<int *>(rinvs + NBLK * NB);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane % MT, ks = lane / MT;
    auto acc_row = [&](int reg) { return (sizeof(T) == 4) ? ((reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) : ((lane >> 4) + 4 * reg); };
    if (tid == 0) *failp = 0;
#ifdef NMFX_POTRF_TIMING
    if (lane == 0) { nmfx_potrf_dbg[128 + wave] = (long long)__builtin_readcyclecounter(); nmfx_potrf_dbg[136 + wave] = (long long)wall_clock64(); }
#endif
    // the blocks of this wave
    int sbi[NS], sbj[NS];
    typename M::acc_t acc[NS][SUB][SUB];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int t = s * NW + wave;
        int bj = 0;
        while ((bj + 1) * (bj + 2) / 2 <= t) ++bj;
        sbj[s] = (t < NT) ? bj : -1;
        sbi[s] = (t < NT) ? t - bj * (bj + 1) / 2 : -1;
        // (every load of the wave in flight before the first use)
#pragma unroll
        for (int si = 0; si < SUB; ++si)
#pragma unroll
            for (int sj = 0; sj < SUB; ++sj)
#pragma unroll
                for (int reg = 0; reg < M::NACC; ++reg) {
                    const int gr = NB * sbi[s] + si * MT + li, gc = NB * sbj[s] + sj * MT + acc_row(reg);
                    const bool in = t < NT && gr < k && gc < k;         // (clamped, unconditional loads: conditional ones are waited for one by one)
                    T v = A[in ? gr + (int64_t)gc * ld : 0];
                    v = in ? v : (T)0;
                    if (gr == gc) v = (gr < k) ? v + lambda : (T)1;      // adddiag!; identity padding beyond k
                    acc[s][si][sj][reg] = v;
                }
    }
    lds_barrier();
#ifdef NMFX_POTRF_TIMING
#define PTR(i) if (lane == 0) nmfx_potrf_dbg[256 + (b * 8 + wave) * 8 + (i)] = (long long)__builtin_readcyclecounter();
#else
#define PTR(i)
#endif
    T myrinv = (T)0;
    int next_inv = 0;      // next diagonal block to invert (same count in every wave)
    for (int b = 0; b < NBLK || (Dinv != nullptr && next_inv < NBLK); ++b) {      // (iterations >= NBLK: only the last inverses)
        const int nrem = NBLK - 1 - b;
        PTR(0)
        // A: block row b as column images
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (sbi[s] == b) {
                T *Eb = E + (size_t)(sbj[s] - b) * NB * LDE;
#pragma unroll
                for (int si = 0; si < SUB; ++si)
#pragma unroll
                    for (int sj = 0; sj < SUB; ++sj)
#pragma unroll
                        for (int reg = 0; reg < M::NACC; ++reg) Eb[(sj * MT + acc_row(reg)) * LDE + si * MT + li] = acc[s][si][sj][reg];
            }
        PTR(1)
        lds_barrier();
        PTR(2)
        // B: elimination of [diagonal block | panel block wave]
        // The inverses of the finished diagonal blocks (Dinv != nullptr) are dealt to waves 0 .. 3 on the SIMDs that have NO elimination wave
        // in this step (wave w runs on SIMD w % 4): two of these instruction-bound chains on one SIMD take 1.5x as long as one (measured:
        // 15 k cycles against 10.5 k), and the elimination is the critical path of the step.  At k = 256 that is from step 4 on:
        // inv(0) | inv(1), inv(2) | inv(3), inv(4), inv(5) | inv(6) | inv(7).
        const int nel = (b < NBLK) ? (nrem > 0 ? nrem : 1) : 0;
        int my_inv = -1;
        if (Dinv != nullptr)
            for (int w = nel; w < 4; ++w)
                if (next_inv < b && next_inv < NBLK) {
                    if (wave == w) my_inv = next_inv;
                    ++next_inv;
                }
        if (wave < nel) {
            const T *src = (lane < NB || nrem == 0) ? E + (size_t)(lane & 31) * LDE : E + (size_t)(1 + wave) * NB * LDE + (size_t)(lane - NB) * LDE;
            T col[NB];
#pragma unroll
            for (int q = 0; q < NB / V; ++q) {
                const vec_t v = *reinterpret_cast<const vec_t *>(src + q * V);
#pragma unroll
                for (int u = 0; u < V; ++u) col[q * V + u] = v[u];
            }
            bool bad = false;
            chol_eliminate32(col, lane, myrinv, bad);
            if (nrem > 0 && lane >= NB) {
                const int c = NB * wave + lane - NB;                      // column of the panel
#pragma unroll
                for (int l = 0; l < NB; ++l) Rp[l * KP + c] = col[l];
            }
            if (wave == 0 && lane < NB) {
                // the factored diagonal block: to LDS as a column image -- copied to global memory in phase C, inverted in a later step
                T *Ub = E + (size_t)(NBLK - b) * NB * LDE;
#pragma unroll
                for (int q = 0; q < NB / V; ++q) {
                    vec_t v;
#pragma unroll
                    for (int u = 0; u < V; ++u) v[u] = col[q * V + u];
                    *reinterpret_cast<vec_t *>(Ub + lane * LDE + q * V) = v;
                }
                rinvs[b * NB + lane] = myrinv;
                if (bad && lane == 0) *failp = 1;
            }
        } else if (my_inv >= 0) {
            // inverse of a finished diagonal block: column `lane` of inv(U_bb) by back-substitution on e_lane (trtri_diag_kernel), the
            // divisions by the diagonal as multiplications by the reciprocals the factorisation formed (as LAPACK's trti2 does)
            const int pb = my_inv;
            T *Ub = E + (size_t)(NBLK - pb) * NB * LDE;
            const T *src = Ub + (size_t)(lane & 31) * LDE;
            T col[NB], v[NB];
#pragma unroll
            for (int q = 0; q < NB / V; ++q) {
                const vec_t x = *reinterpret_cast<const vec_t *>(src + q * V);
#pragma unroll
                for (int u = 0; u < V; ++u) col[q * V + u] = x[u];
            }
            const T ri = rinvs[pb * NB + (lane & 31)];
            strip_static_for<NB>([&](auto ic) {
                constexpr int i = NB - 1 - decltype(ic)::value;
                // (every U(i, l) is known up front: left alone, the compiler broadcasts all 496 of them ahead of the recurrence and spills the
                // scalars; tying row i to the result of row i + 1 keeps each broadcast next to its use -- and the 32 masks out of the scalars)
                int lj = lane & 31;
                T ci = col[i];
                if constexpr (i < NB - 1) pin_v(lj, ci, v[i + 1 < NB ? i + 1 : i]);
                else pin_v(lj, ci);
                T s = (lj == i) ? (T)1 : (T)0;
                strip_static_for<(NB - i - 1 + 7) / 8>([&](auto gc_) {      // (broadcasts eight at a time ahead of their uses, as in the elimination)
                    constexpr int l0 = i + 1 + 8 * decltype(gc_)::value;
                    T m[8];
                    strip_static_for<8>([&](auto uc) {
                        constexpr int u = decltype(uc)::value;
                        if constexpr (l0 + u < NB) m[u] = lane_bcast(ci, l0 + u); else m[u] = (T)0;
                    });
                    asm volatile("" : "+s"(m[0]), "+s"(m[1]), "+s"(m[2]), "+s"(m[3]), "+s"(m[4]), "+s"(m[5]), "+s"(m[6]), "+s"(m[7]));
                    strip_static_for<8>([&](auto uc) {
                        constexpr int u = decltype(uc)::value;
                        if constexpr (l0 + u < NB) s = nmfx_fma(-m[u], v[l0 + u], s);
                    });
                });
                v[i] = s * lane_bcast(ri, i);
            });
            // out through the block's own LDS image (this wave is its last reader), then with a run-time triangle test (32 compile-time
            // masks r <= lane end up hoisted out of the block loop and spilled); the 16 LDS reads first, then the 16 stores
            if (lane < NB) {
#pragma unroll
                for (int q = 0; q < NB / V; ++q) {
                    vec_t x;
#pragma unroll
                    for (int u = 0; u < V; ++u) x[u] = v[q * V + u];
                    *reinterpret_cast<vec_t *>(Ub + lane * LDE + q * V) = x;
                }
            }
            store_upper32<T, LDE>(Dinv + (int64_t)NB * pb + (int64_t)NB * pb * ld, ld, Ub, k - NB * pb, lane);
        }
        PTR(3)
        lds_barrier();
        PTR(4)
        if (*failp) break;
        if (wave == NW - 1 && b < NBLK) {      // the factored diagonal block to global memory (upper triangle), off the elimination waves
            store_upper32<T, LDE>(A + (int64_t)NB * b + (int64_t)NB * b * ld, ld, E + (size_t)(NBLK - b) * NB * LDE, k - NB * b, lane);
            PTR(6)
        }
        // ... and the solved row panel, from its LDS image, 16 bytes per lane with 32 / V lanes per column: an instruction writes 2 V whole
        // columns of the block row.  (Stored from the elimination's registers -- lane = column -- every instruction touched 32 lines, and
        // the CU's store path took 5-6 k cycles per step to drain them; as 4-byte stores there were 4x as many instructions.)
        if (b < NBLK) {
            constexpr int LPC = NB / V, CPI = 64 / LPC, NIT = (NB * (NBLK - 1) / CPI + NW - 1) / NW;
            const int q = lane % LPC, cc = lane / LPC;
            vec_t pv[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = CPI * (wave + NW * it) + cc;
#pragma unroll
                for (int u = 0; u < V; ++u) pv[it][u] = (c < NB * nrem) ? Rp[(V * q + u) * KP + c] : (T)0;
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = CPI * (wave + NW * it) + cc, gc = NB * (b + 1) + c;
                if (c < NB * nrem && gc < k) *reinterpret_cast<vec_t *>(A + (NB * b + V * q) + (int64_t)gc * ld) = pv[it];
            }
        }
        // C: trailing update of the blocks below block row b
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (sbi[s] > b && b < NBLK) {
                const T *pa = Rp + (size_t)NB * (sbj[s] - b - 1) + li, *pb_ = Rp + (size_t)NB * (sbi[s] - b - 1) + li;
#pragma unroll
                for (int kk = 0; kk < NB / KS; ++kk) {
                    T af[SUB], bf[SUB];
#pragma unroll
                    for (int q = 0; q < SUB; ++q) {
                        af[q] = -pa[(kk * KS + ks) * KP + q * MT];      // D rows (registers) <-> matrix column
                        bf[q] = pb_[(kk * KS + ks) * KP + q * MT];      // D columns (lanes)  <-> matrix row
                    }
#pragma unroll
                    for (int si = 0; si < SUB; ++si)
#pragma unroll
                        for (int sj = 0; sj < SUB; ++sj) acc[s][si][sj] = M::mma(af[sj], bf[si], acc[s][si][sj]);
                }
            }
        PTR(5)
    }
    if (*failp) {
        if (tid == 0 && ctrl != nullptr) {
            ctrl->status = posdef_status;
            ctrl->done = 1;
        }
        return;
    }
}

// ---------------------------------------------------------------------------------------------
// Triangular inverse Uinv = inv(U) (what potri!'s first half, trtri, computes), blocked by 32:
//   Uinv[b,b] = inv(U[b,b])                                              -- trtri_diag_kernel
//   Uinv[a,b] = -Uinv[a,a] * sum_{c=a+1..b} U[a,c] * Uinv[c,b],  a < b    -- trtri_offdiag_kernel
// Block columns b are independent (one workgroup each); inside one, the block rows are a serial chain
// a = b-1 .. 0 of 32 x 32 x 32 products on the matrix cores, with the finished tiles of the column kept in LDS.
// ---------------------------------------------------------------------------------------------

// grid = ceil(k/32) workgroups of one wave.  Lane c computes column c of inv(U_bb) by back-substitution on e_c; the
// needed U(i,l) live in lane l's registers and are broadcast with v_readlane (branch-free, identity-padded).
template <typename T>
__global__ __launch_bounds__(64) void trtri_diag_kernel(const T *U, T *Uinv, int64_t ld, int k, const int *done) {
    NMFX_DONE_GUARD(done);
    __builtin_amdgcn_s_setprio(3);
    constexpr int NB = 32;
    const int jb = blockIdx.x * NB, lane = threadIdx.x;
    const int nb = (k - jb < NB) ? (k - jb) : NB;
    T col[NB], v[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) {
        const bool in = (lane < nb) && (r <= lane);
        col[r] = in ? U[(jb + r) + (int64_t)(jb + (in ? lane : 0)) * ld] : ((r == lane) ? (T)1 : (T)0);
    }
#pragma unroll
    for (int i = NB - 1; i >= 0; --i) {
        // v[i] = (delta_{i,lane} - sum_{l=i+1..31} U(i,l) v[l]) / U(i,i);   U(i,l) = lane l's col[i] (zero for l < i)
        T s = (lane == i) ? (T)1 : (T)0;
#pragma unroll
        for (int l = i + 1; l < NB; ++l) s -= lane_bcast(col[i], l) * v[l];
        v[i] = s / lane_bcast(col[i], i);
    }
#pragma unroll
    for (int r = 0; r < NB; ++r)
        if (lane < nb && r <= lane) Uinv[(jb + r) + (int64_t)(jb + lane) * ld] = v[r];
}

// grid = ceil(k/32) workgroups (block column b) of 4 waves.  LDS: the `nfit` most recently finished tiles Y[c] of the column
// (slot b - c, row-major 32x32 each; the diagonal tile is slot 0) + 4 partial-sum tiles: dynamic LDS = (min(nblk, nfit) + 4) * 1024
// elements of T.  Tiles further down the chain than `nfit` are read back from Uinv in global memory (written by this workgroup, L2
// hits): no limit on k from the LDS size (a fully LDS-resident column needs 160 KiB at k = 512 in f64).
template <typename T>
__global__ __launch_bounds__(256) void trtri_offdiag_kernel(const T *U, T *Uinv, int64_t ld, int k, int nfit, const int *done) {
    NMFX_DONE_GUARD(done);
    __builtin_amdgcn_s_setprio(3);
    using M = Mfma<T>;
    constexpr int NB = 32, MT = M::MT, KS = M::KS, SUB = NB / MT;   // SUB x SUB MFMA tiles per 32 x 32 block
    extern __shared__ __attribute__((aligned(16))) unsigned char chol_smem[];
    T *Y = reinterpret_cast<T *>(chol_smem);          // Y[(b-c)*1024 + l*32 + j] = Uinv(32c + l, 32b + j), b - c < nfit
    const int nblk = (k + NB - 1) / NB;
    T *Ps = Y + (size_t)((nblk < nfit) ? nblk : nfit) * NB * NB;   // 4 partial tiles, same row-major layout
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane % MT, ks = lane / MT;
    // Y[b] = diagonal block (already in Uinv), zero-padded
    for (int e = tid; e < NB * NB; e += 256) {
        const int l = e / NB, j = e % NB;
        const int gr = b * NB + l, gc = b * NB + j;
        Y[e] = (gr < k && gc < k && l <= j) ? Uinv[gr + (int64_t)gc * ld] : (T)0;
    }
    __syncthreads();
    for (int a = b - 1; a >= 0; --a) {
        // partial sums: wave w adds the products U[a,c] * Y[c] for c = a+1+w, a+1+w+4, ...
        typename M::acc_t acc[SUB][SUB];
#pragma unroll
        for (int si = 0; si < SUB; ++si)
#pragma unroll
            for (int sj = 0; sj < SUB; ++sj)
#pragma unroll
                for (int r = 0; r < M::NACC; ++r) acc[si][sj][r] = (T)0;
        // (round 6: the operand loads of a block product are requested together, ahead of its 16 matrix-core steps -- they used to sit
        // inside the k-loop behind a run-time `tile in LDS?` branch, i.e. one global round trip per k-step: 12.5 us per block step, 88 us
        // for the launch at k = 256; and wave 0 requests the diagonal block of its closing product here, at the top of the step)
        T vinv[NB / KS][SUB];
        if (wave == 0) {
#pragma unroll
            for (int kk = 0; kk < NB / KS; ++kk)
#pragma unroll
                for (int si = 0; si < SUB; ++si) {     // A(i, l) = Vinv_a(i, l), upper triangular
                    const int i = si * MT + li, l = kk * KS + ks;
                    const int gr = a * NB + i, gc = a * NB + l;
                    const bool in = i <= l && gr < k && gc < k;
                    const T v = Uinv[in ? gr + (int64_t)gc * ld : 0];
                    vinv[kk][si] = in ? v : (T)0;
                }
        }
        for (int c = a + 1 + wave; c <= b; c += 4) {
            const bool in_lds = (b - c) < nfit;
            const T *Yc = Y + (size_t)(in_lds ? (b - c) : 0) * NB * NB;
            T afr[NB / KS][SUB];
#pragma unroll
            for (int kk = 0; kk < NB / KS; ++kk)
#pragma unroll
                for (int si = 0; si < SUB; ++si) {     // A(i, l) = U(32a + i, 32c + l)
                    const int gr = a * NB + si * MT + li, gc = c * NB + kk * KS + ks;
                    const bool in = gr < k && gc < k;
                    const T v = U[in ? gr + (int64_t)gc * ld : 0];
                    afr[kk][si] = in ? v : (T)0;
                }
            auto run = [&](auto in_lds_c) {
                constexpr bool IN_LDS = decltype(in_lds_c)::value;
                T bfr[NB / KS][SUB];
#pragma unroll
                for (int kk = 0; kk < NB / KS; ++kk)
#pragma unroll
                    for (int sj = 0; sj < SUB; ++sj) {
                        const int l = kk * KS + ks;
                        if constexpr (IN_LDS) bfr[kk][sj] = Yc[l * NB + sj * MT + li];
                        else {   // finished tile c of this column, from global memory: Uinv(32c + l, 32b + j); zero outside the matrix
                            const int gr = c * NB + l, gc = b * NB + sj * MT + li;
                            const bool in = gr < k && gc < k;
                            const T v = *reinterpret_cast<const volatile T *>(Uinv + (in ? gr + (int64_t)gc * ld : 0));
                            bfr[kk][sj] = in ? v : (T)0;
                        }
                    }
#pragma unroll
                for (int kk = 0; kk < NB / KS; ++kk)
#pragma unroll
                    for (int si = 0; si < SUB; ++si)
#pragma unroll
                        for (int sj = 0; sj < SUB; ++sj) acc[si][sj] = M::mma(afr[kk][si], bfr[kk][sj], acc[si][sj]);
            };
            if (in_lds) run(std::true_type{});
            else run(std::false_type{});
        }
        // write this wave's partial tile (row-major) and combine the four in a fixed order
        T *Pw = Ps + (size_t)wave * NB * NB;
#pragma unroll
        for (int si = 0; si < SUB; ++si)
#pragma unroll
            for (int sj = 0; sj < SUB; ++sj)
#pragma unroll
                for (int reg = 0; reg < M::NACC; ++reg) {
                    int rr;
                    if constexpr (sizeof(T) == 4) rr = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                    else rr = (lane >> 4) + 4 * reg;
                    Pw[(si * MT + rr) * NB + sj * MT + li] = acc[si][sj][reg];
                }
        __syncthreads();
        for (int e = tid; e < NB * NB; e += 256)
            Ps[e] = ((Ps[e] + Ps[NB * NB + e]) + Ps[2 * NB * NB + e]) + Ps[3 * NB * NB + e];
        __syncthreads();
        // Y[a] = -Vinv_a * S   (wave 0; Vinv_a = Uinv[a,a] from global, S = Ps[0])
        if (wave == 0) {
            typename M::acc_t acc2[SUB][SUB];
#pragma unroll
            for (int si = 0; si < SUB; ++si)
#pragma unroll
                for (int sj = 0; sj < SUB; ++sj)
#pragma unroll
                    for (int r = 0; r < M::NACC; ++r) acc2[si][sj][r] = (T)0;
#pragma unroll
            for (int kk = 0; kk < NB / KS; ++kk) {
                const int l = kk * KS + ks;
                T bf[SUB];
#pragma unroll
                for (int sj = 0; sj < SUB; ++sj) bf[sj] = Ps[l * NB + sj * MT + li];
#pragma unroll
                for (int si = 0; si < SUB; ++si)
#pragma unroll
                    for (int sj = 0; sj < SUB; ++sj) acc2[si][sj] = M::mma(vinv[kk][si], bf[sj], acc2[si][sj]);
            }
            const bool keep = (b - a) < nfit;
            T *Ya = Y + (size_t)(keep ? (b - a) : 0) * NB * NB;
#pragma unroll
            for (int si = 0; si < SUB; ++si)
#pragma unroll
                for (int sj = 0; sj < SUB; ++sj)
#pragma unroll
                    for (int reg = 0; reg < M::NACC; ++reg) {
                        int rr;
                        if constexpr (sizeof(T) == 4) rr = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                        else rr = (lane >> 4) + 4 * reg;
                        const int i = si * MT + rr, j = sj * MT + li;
                        const T val = -acc2[si][sj][reg];
                        if (keep) Ya[i * NB + j] = val;
                        const int gr = a * NB + i, gc = b * NB + j;
                        if (gr < k && gc < k) Uinv[gr + (int64_t)gc * ld] = val;
                    }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// potrs!: X = inv(U'U) B by two triangular substitutions (pdsolve!, src/utils.jl:63-70: potrf! + potrs!) -- the reference's route,
// blocked by 32:   U'y = b :  y_i = inv(U_ii)' (b_i - sum_{j<i} U_ji' y_j)      x from U x = y :  x_i = inv(U_ii) (y_i - sum_{j>i} U_ij x_j)
// Half the flops of the product form Uinv (Uinv' B) (rounds 1-3) and no triangular inverse beyond the 32 x 32 diagonal blocks.
//
// potrs_prep_kernel packs what both sweeps read into ONE k x k matrix Tm (leading dimension ld, zero outside k x k):
//   off-diagonal blocks : upper = U, lower = U' (so that BOTH sweeps read the block they need with the row index contiguous)
//   diagonal blocks     : Dinv_i = inv(U_ii) in the upper triangle (diagonal included), its transpose in the strictly lower one
// Every product of either sweep is then  acc(a, c) = sum_l Tm(32 j + a, 32 i + l) * S(32 i + l, c)  -- the trailing updates with
// j != i (forward: j > i, backward: j < i), the diagonal solves with j == i and the other triangle masked.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void potrs_prep_kernel(const T *U, const T *Dinv, T *Tm, int64_t ld, int k, int K, const int *done) {
    NMFX_DONE_GUARD(done);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (int64_t)K * K; e += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(e % K), c = (int)(e / K);
        T v = (T)0;
        if (r < k && c < k) {
            if ((r >> 5) == (c >> 5)) v = (r <= c) ? Dinv[r + (int64_t)c * ld] : Dinv[c + (int64_t)r * ld];
            else v = (r < c) ? U[r + (int64_t)c * ld] : U[c + (int64_t)r * ld];
        }
        Tm[r + (int64_t)c * ld] = v;
    }
}

// One workgroup (8 waves) per panel of NB columns of B, the K x NB panel resident in LDS for both sweeps (S[row][col], row stride
// NB + 1).  Right-looking: block step i solves its 32 rows against the diagonal block (a wave owns all 32 rows of its 32 / 16
// columns, so the solve is in place without a barrier), then ALL later block rows take their update  r_j -= Tm_ji y_i  on the matrix
// cores, (block row, column tile) items dealt round-robin to the waves.  Block column i of Tm (K x 32: every operand of step i) is
// staged through LDS as TP[l][row] -- coalesced 16-byte loads, fragment reads with the lanes on consecutive rows -- and the NEXT
// step's block column is requested into registers at the start of a step, so its L2 round trip runs under the step's products
// (the first version read its fragments straight from global memory: 114 us at 16384 columns, k = 256, a dependent L2 round trip
// per item; the product form it replaces took 55 us).  DBUF: two TP buffers (one barrier less per step) when the LDS holds them.
// Epilogue: projectnn! (max(x, 0), NaN passes through, src/utils.jl:34-41) if clamp, store; with `old` != nullptr also
// stop_condition's sums of every component over the panel's columns (src/common.jl:100-104), partial[(panel * ncomp + a) * 2 + {0, 1}]
// -- the layout finalize_partials_kernel reduces.  B = sum of nslab slabs (ascending), like the split-K combine it replaces.
template <typename T, int NB, int POTRS_THREADS>
__global__ __launch_bounds__(POTRS_THREADS) void potrs_panel_kernel(const T *Tm, int64_t ldt, const T *B, int nslab, int64_t slab_stride, int64_t ldb, T *Xout,
                                                                    const T *old, int K, int clamp, int dbuf, double *stat_partial, int ncomp, const int *done) {
    NMFX_DONE_GUARD(done);
    using M = Mfma<T>;
    constexpr int MT = M::MT, KS = M::KS, SUB = 32 / MT, NTN = NB / MT, LDP = NB + 1, NW = POTRS_THREADS / 64, V = 16 / (int)sizeof(T);
    constexpr int MAXQ = 8;   // 16-byte chunks of a block column per thread: K * 32 / V / THREADS <= 8, checked by the host
    using vec_t = typename M::vec_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char chol_smem[];
    T *TP = reinterpret_cast<T *>(chol_smem);                 // [dbuf ? 2 : 1][32][K]
    T *S = TP + (size_t)(dbuf ? 2 : 1) * 32 * K;              // [K][LDP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane % MT, ks = lane / MT;
    const int64_t c0 = (int64_t)blockIdx.x * NB;
    const int nb = K / 32, nsteps = 2 * nb;
    const int nq = K * 32 / V / POTRS_THREADS;                // chunks per thread (K is a multiple of 64)
    auto step_col = [&](int s) { return s < nb ? s : 2 * nb - 1 - s; };
    vec_t pre[MAXQ];
    auto tp_load = [&](int s) {                               // block column of step s -> registers (chunk c: column l = c / (K / V), rows V * (c % (K / V)) ..)
        const int i = step_col(s);
#pragma unroll
        for (int q = 0; q < MAXQ; ++q)
            if (q < nq) {
                const int c = tid + POTRS_THREADS * q, l = c / (K / V), r = V * (c % (K / V));
                pre[q] = *reinterpret_cast<const vec_t *>(Tm + r + (int64_t)(32 * i + l) * ldt);
            }
    };
    auto tp_store = [&](int buf) {
#pragma unroll
        for (int q = 0; q < MAXQ; ++q)
            if (q < nq) {
                const int c = tid + POTRS_THREADS * q, l = c / (K / V), r = V * (c % (K / V));
                *reinterpret_cast<vec_t *>(TP + (size_t)buf * 32 * K + l * K + r) = pre[q];
            }
    };
    tp_load(0);
    // panel in: column c of B is contiguous in the component a
    for (int e0 = tid; e0 < K * NB; e0 += 8 * POTRS_THREADS) {     // 8 independent loads per trip
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * POTRS_THREADS;
            v[u] = (e < K * NB) ? B[e % K + (c0 + e / K) * ldb] : (T)0;
        }
        for (int q = 1; q < nslab; ++q)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * POTRS_THREADS;
                if (e < K * NB) v[u] += B[(int64_t)q * slab_stride + e % K + (c0 + e / K) * ldb];
            }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * POTRS_THREADS;
            if (e < K * NB) S[(e % K) * LDP + e / K] = v[u];
        }
    }
    tp_store(0);
    __syncthreads();
    auto acc_row = [&](int reg) { return (sizeof(T) == 4) ? ((reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) : ((lane >> 4) + 4 * reg); };
    // acc[si] (rows 32 jb + si*MT .., columns nj*MT ..) = sum_l Tm(32 jb + a, 32 ib + l) S(32 ib + l, c);  tri: 0 none, +1 keep a >= l, -1 keep a <= l
    auto product = [&](typename M::acc_t (&acc)[SUB], const T *tp, int jb, int ib, int nj, int tri) {
#pragma unroll
        for (int si = 0; si < SUB; ++si)
#pragma unroll
            for (int r = 0; r < M::NACC; ++r) acc[si][r] = (T)0;
        T af[32 / KS][SUB], bf[32 / KS];
#pragma unroll
        for (int kk = 0; kk < 32 / KS; ++kk) {
            const int l = kk * KS + ks;
#pragma unroll
            for (int si = 0; si < SUB; ++si) {
                const int a = si * MT + li;
                T v = tp[l * K + 32 * jb + a];
                if (tri > 0) v = (a >= l) ? v : (T)0;
                if (tri < 0) v = (a <= l) ? v : (T)0;
                af[kk][si] = v;
            }
            bf[kk] = S[(32 * ib + l) * LDP + nj * MT + li];
        }
#pragma unroll
        for (int kk = 0; kk < 32 / KS; ++kk)
#pragma unroll
            for (int si = 0; si < SUB; ++si) acc[si] = M::mma(af[kk][si], bf[kk], acc[si]);
    };
    for (int s = 0; s < nsteps; ++s) {
        const int sweep = s < nb ? 0 : 1, step = s < nb ? s : s - nb, i = step_col(s);
        const int buf = dbuf ? (s & 1) : 0;
        const T *tp = TP + (size_t)buf * 32 * K;
        if (s + 1 < nsteps) tp_load(s + 1);
        // diagonal solve, in place: wave w owns column tiles w, w + NW, ...
        for (int nj = wave; nj < NTN; nj += NW) {
            typename M::acc_t acc[SUB];
            product(acc, tp, i, i, nj, sweep == 0 ? 1 : -1);
#pragma unroll
            for (int si = 0; si < SUB; ++si)
#pragma unroll
                for (int reg = 0; reg < M::NACC; ++reg) S[(32 * i + si * MT + acc_row(reg)) * LDP + nj * MT + li] = acc[si][reg];
        }
        __syncthreads();
        // trailing update of every block row still to be solved in this sweep
        const int nrem = nb - 1 - step;
        for (int it = wave; it < nrem * NTN; it += NW) {
            const int jr = it / NTN, nj = it % NTN;
            const int j = sweep == 0 ? i + 1 + jr : i - 1 - jr;
            typename M::acc_t acc[SUB];
            product(acc, tp, j, i, nj, 0);
#pragma unroll
            for (int si = 0; si < SUB; ++si)
#pragma unroll
                for (int reg = 0; reg < M::NACC; ++reg) S[(32 * j + si * MT + acc_row(reg)) * LDP + nj * MT + li] -= acc[si][reg];
        }
        if (s + 1 < nsteps) {
            if (!dbuf) __syncthreads();                       // single buffer: every fragment read of this step first
            tp_store(dbuf ? (buf ^ 1) : 0);
        }
        __syncthreads();
    }
    // panel out: thread = component; 8 columns per trip so that the loads of `old` are in flight together (one column per trip
    // serialised a global round trip per column behind the store of the previous one: 38 of the first version's 103 us)
    for (int a = tid; a < K; a += POTRS_THREADS) {
        double dev = 0.0, sum = 0.0;
        for (int cb = 0; cb < NB; cb += 8) {
            T ov[8], v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) ov[u] = (old != nullptr) ? old[a + (c0 + cb + u) * ldb] : (T)0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v[u] = S[a * LDP + cb + u];
                if (clamp) v[u] = (v[u] < (T)0) ? (T)0 : v[u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) Xout[a + (c0 + cb + u) * ldb] = v[u];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const T d = v[u] - ov[u], sp = v[u] + ov[u];
                dev += (double)(T)(d * d);
                sum += (double)(T)(sp * sp);
            }
        }
        if (old != nullptr && stat_partial != nullptr) {
            stat_partial[((int64_t)blockIdx.x * ncomp + a) * 2] = dev;
            stat_partial[((int64_t)blockIdx.x * ncomp + a) * 2 + 1] = sum;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// potrs! by STRIPS (round 5): the two substitutions of src/utils.jl:69 with no LDS panel, no barrier and no cross-wave traffic.
//
// The columns of the right-hand side are independent, so ONE WAVE owns a strip of 16 columns for both sweeps and keeps the whole
// K x 16 strip in accumulator registers: block row i (32 rows) is two 16 x 16 tiles of v_mfma_{f32,f64}_16x16x4, i.e. 2 x 4
// registers per lane.  Left-looking:  r_i = b_i - sum_{j < i} L_ij y_j  (the matrix core accumulates into b_i; the packed blocks are
// stored NEGATED), then  y_i = inv(L_ii) r_i  as one more block product; the backward sweep mirrors it with U.  The trick that makes
// the registers enough: the B operand of the 16x16x4 instruction wants lane (n, g) to supply contraction index (k-step, g), and an
// accumulator tile holds rows {4 g + r} (Float32) / {g + 4 r} (Float64) of column n in lane (n, g) -- so if k-step kk of a block
// product is DEFINED to contract row  rho(kk, g) = 16 (kk / 4) + 4 g + kk % 4  (Float32;  16 (kk / 4) + g + 4 (kk % 4)  in Float64) of
// the block, the B operand IS accumulator register kk % 4 of tile kk / 4: finished block rows feed the next products straight from
// the registers the matrix core wrote them to (any partition of the contraction is a valid order of the sum; cd.hpp uses the same
// freedom).  The A operand -- the 32 x 32 blocks of the factor, the same for every strip -- is packed once per factorisation in
// exactly the order the sweeps consume it (potrs_strip_pack_kernel): block after block, inside a block 16 values per lane as
// consecutive 16-byte chunks, so a wave's load instruction is one contiguous KiB and the walk over the factor is a linear stream
// from L2, requested TWO blocks ahead (Float64: one).  A panel's chain of 2 K / 32 barrier-separated steps (potrs_panel_kernel: 79 us at
// 16384 columns, k = 256, Float32; the product form Uinv (Uinv' B): 55 us) becomes 72 back-to-back block products per wave.
// NBLK = K / 32 at compile time (every loop is unrolled: the strip's registers are indexed statically): 2, 4, 6, 8, and in Float32 -- where
// the strip of K = 512 is 128 registers of a one-wave-per-SIMD budget of 512 -- also 10, 12, 14, 16.
// ---------------------------------------------------------------------------------------------
template <typename T> __host__ __device__ inline int strip_rho(int kk, int g) {
    return sizeof(T) == 4 ? 16 * (kk >> 2) + 4 * g + (kk & 3) : 16 * (kk >> 2) + g + 4 * (kk & 3);
}
// blocks of one sweep in consumption order: step s = 0 .. nb - 1 takes s off-diagonal blocks, then its diagonal block
__host__ __device__ constexpr int strip_step_of(int b) { int s = 0; while ((s + 1) * (s + 2) / 2 <= b) ++s; return s; }
__host__ __device__ constexpr int64_t strip_pack_elems(int nb) { return (int64_t)nb * (nb + 1) * 1024; }

// Tp[block][chunk q][lane][v]: value m = q * V + v of lane (n, g) is  A(16 (m / 8) + n, rho(m % 8, g))  of the block, where the block is
//   forward  step i, j < i : -L_ij = -U_ji'      diagonal: inv(L_ii) = inv(U_ii)' (lower triangle)
//   backward step s (i = nb - 1 - s), j = i + 1 + jpos : -U_ij      diagonal: inv(U_ii) (upper triangle)
// U: the factor (upper triangle of A after potrf), Dinv: inv(U_ii) in the upper triangles of the diagonal blocks (trtri_diag_kernel).
template <typename T>
__global__ void potrs_strip_pack_kernel(const T *U, const T *Dinv, T *Tp, int64_t ld, int k, int nb, const int *done) {
    NMFX_DONE_GUARD(done);
    constexpr int V = 16 / (int)sizeof(T);
    const int nf = nb * (nb + 1) / 2;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < strip_pack_elems(nb); e += (int64_t)gridDim.x * blockDim.x) {
        const int blk = (int)(e >> 10), w = (int)(e & 1023);
        const int q = w / (64 * V), lane = (w / V) & 63, v = w % V, m = q * V + v;
        const int a = 16 * (m >> 3) + (lane & 15), l = strip_rho<T>(m & 7, lane >> 4);
        const bool back = blk >= nf;
        const int b = back ? blk - nf : blk;
        int st = 0;
        while ((st + 1) * (st + 2) / 2 <= b) ++st;
        const int jpos = b - st * (st + 1) / 2;
        const bool diag = jpos == st;
        const int i = back ? nb - 1 - st : st;
        const int j = diag ? i : (back ? i + 1 + jpos : jpos);
        const int r = 32 * i + a, c = 32 * j + l;      // element (r, c) of the K x K operator applied from the left
        T val = (T)0;
        if (r < k && c < k) {
            if (diag) {
                if (!back && a >= l) val = Dinv[c + (int64_t)r * ld];        // inv(L_ii)(a, l) = inv(U_ii)(l, a)
                if (back && a <= l) val = Dinv[r + (int64_t)c * ld];
            } else {
                val = back ? -U[r + (int64_t)c * ld] : -U[c + (int64_t)r * ld];   // forward: L(r, c) = U(c, r)
            }
        }
        Tp[e] = val;
    }
}

template <typename T> struct StripMfma;
template <> struct StripMfma<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    typedef float vec_t __attribute__((ext_vector_type(4)));
    static constexpr int V = 4, NQ = 4, AHEAD = 2;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
};
template <> struct StripMfma<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    typedef double vec_t __attribute__((ext_vector_type(2)));
    static constexpr int V = 2, NQ = 8, AHEAD = 1;
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
};

constexpr int STRIP_WAVES = 4, STRIP_COLS = 16 * STRIP_WAVES;

// B = sum of nslab slabs (ascending); epilogue as potrs_panel_kernel: projectnn! if clamp, store, and with `old` stop_condition's sums of
// every component over the workgroup's 64 columns: stat_partial[(blockIdx.x * ncomp + a) * 2 + {0, 1}] (finalize_partials_kernel's layout).
template <typename T, int NBLK>
__global__ __launch_bounds__(64 * STRIP_WAVES) void potrs_strip_kernel(const T *Tp, const T *B, int nslab, int64_t slab_stride, int64_t ldb, T *Xout, const T *old,
                                                                       int clamp, double *stat_partial, int ncomp, const int *done) {
    NMFX_DONE_GUARD(done);
    using M = StripMfma<T>;
    using acc_t = typename M::acc_t;
    using vec_t = typename M::vec_t;
    constexpr int V = M::V, NQ = M::NQ, AHEAD = M::AHEAD, RING = AHEAD + 1, NF = NBLK * (NBLK + 1) / 2, NTOT = 2 * NF;
    __shared__ double red[STRIP_WAVES][NBLK * 32][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    const int64_t col = ((int64_t)blockIdx.x * STRIP_WAVES + wave) * 16 + n;
    // row of register r of tile (i, si) in this lane
    auto row_of = [&](int i, int si, int r) { return 32 * i + 16 * si + (sizeof(T) == 4 ? 4 * g + r : g + 4 * r); };
    vec_t frag[RING][NQ];
    auto request = [&](int b, int slot) {
        const T *src = Tp + (int64_t)b * 1024 + lane * V;
#pragma unroll
        for (int q = 0; q < NQ; ++q) frag[slot][q] = *reinterpret_cast<const vec_t *>(src + q * 64 * V);
    };
    strip_static_for<AHEAD>([&](auto b) { request(b, b % RING); });
    // the strip: Y[i][si] = rows of block row i, tile si
    acc_t Y[NBLK][2];
    auto load_strip = [&](acc_t (&dst)[NBLK][2], const T *src0) {      // every tile's load in flight at once
        strip_static_for<NBLK>([&](auto i) {
#pragma unroll
            for (int si = 0; si < 2; ++si) {
                if constexpr (sizeof(T) == 4) {
                    dst[i][si] = *reinterpret_cast<const acc_t *>(src0 + row_of(i, si, 0) + col * ldb);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[i][si][r] = src0[row_of(i, si, r) + col * ldb];
                }
            }
        });
    };
    load_strip(Y, B);
    for (int q = 1; q < nslab; ++q) {
        acc_t Z[NBLK][2];
        load_strip(Z, B + (int64_t)q * slab_stride);
        strip_static_for<NBLK>([&](auto i) { Y[i][0] += Z[i][0]; Y[i][1] += Z[i][1]; });
    }
    // block product: acc[si] += A(block b) * Y[j]   (B operand of k-step kk = register kk % 4 of tile kk / 4 of Y[j])
    auto product = [&](acc_t (&acc)[2], int slot, const acc_t (&src)[2]) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int si = 0; si < 2; ++si) {
                const int m = si * 8 + kk;
                acc[si] = M::mma(frag[slot][m / V][m % V], src[kk >> 2][kk & 3], acc[si]);
            }
    };
    acc_t OV[NBLK][2];
    strip_static_for<NBLK>([&](auto i) { OV[i][0] = OV[i][1] = acc_t{(T)0, (T)0, (T)0, (T)0}; });
    strip_static_for<NTOT>([&](auto bb) {
        constexpr int b = bb, back = b >= NF ? 1 : 0, bs = back ? b - NF : b;
        constexpr int st = strip_step_of(bs), jpos = bs - st * (st + 1) / 2, i = back ? NBLK - 1 - st : st;
        constexpr bool diag = jpos == st;
        constexpr int j = diag ? i : (back ? i + 1 + jpos : jpos);
        if constexpr (b + AHEAD < NTOT) request(b + AHEAD, (b + AHEAD) % RING);
        if constexpr (b == NF) {                                        // `old` for the epilogue: its HBM round trip runs under the backward sweep
            if (old != nullptr) load_strip(OV, old);
        }
        __builtin_amdgcn_sched_barrier(0);                              // (the scheduler otherwise sinks the requests down to their first use)
        if constexpr (diag) {
            acc_t t[2] = {acc_t{(T)0, (T)0, (T)0, (T)0}, acc_t{(T)0, (T)0, (T)0, (T)0}};
            product(t, b % RING, Y[i]);
            Y[i][0] = t[0];
            Y[i][1] = t[1];
        } else {
            product(Y[i], b % RING, Y[j]);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    // epilogue, block row by block row: clamp, store, and stop_condition's terms summed over the strip's 16 columns (the lanes n of a group g)
    // by a halving butterfly -- after the step with mask m a lane keeps the half of its values selected by bit m of n, so the 8 values of a
    // block row cost 4 + 2 + 1 exchanges plus one full exchange instead of 8 x 4; lanes n and n ^ 1 end with the total of value n >> 1
    const bool stats = old != nullptr && stat_partial != nullptr;
    strip_static_for<NBLK>([&](auto i) {
        double dev[8], sum[8];
#pragma unroll
        for (int si = 0; si < 2; ++si) {
            acc_t v = Y[i][si];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (clamp) v[r] = (v[r] < (T)0) ? (T)0 : v[r];
            const acc_t ov = OV[i][si];
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<acc_t *>(Xout + row_of(i, si, 0) + col * ldb) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) Xout[row_of(i, si, r) + col * ldb] = v[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const T d = v[r] - ov[r], sp = v[r] + ov[r];
                dev[si * 4 + r] = (double)(T)(d * d);
                sum[si * 4 + r] = (double)(T)(sp * sp);
            }
        }
        if (stats) {
            auto halve = [&](double (&x)[8], auto cnt_c, int mask) {
                constexpr int cnt = decltype(cnt_c)::value;
                const bool up = (n & mask) != 0;
#pragma unroll
                for (int e = 0; e < cnt / 2; ++e) {
                    const double send = up ? x[e] : x[e + cnt / 2], keep = up ? x[e + cnt / 2] : x[e];
                    x[e] = keep + __shfl_xor(send, mask, 64);
                }
            };
            halve(dev, std::integral_constant<int, 8>{}, 8); halve(dev, std::integral_constant<int, 4>{}, 4); halve(dev, std::integral_constant<int, 2>{}, 2);
            halve(sum, std::integral_constant<int, 8>{}, 8); halve(sum, std::integral_constant<int, 4>{}, 4); halve(sum, std::integral_constant<int, 2>{}, 2);
            const double d1 = dev[0] + __shfl_xor(dev[0], 1, 64), s1 = sum[0] + __shfl_xor(sum[0], 1, 64);
            if ((n & 1) == 0) {
                const int vi = n >> 1, a = row_of(i, vi >> 2, vi & 3);
                red[wave][a][0] = d1;
                red[wave][a][1] = s1;
            }
        }
    });
    if (!stats) return;
    __syncthreads();
    for (int a = threadIdx.x; a < NBLK * 32; a += blockDim.x) {
        double d = 0.0, s2 = 0.0;
#pragma unroll
        for (int w = 0; w < STRIP_WAVES; ++w) { d += red[w][a][0]; s2 += red[w][a][1]; }
        if (a < ncomp) {
            stat_partial[((int64_t)blockIdx.x * ncomp + a) * 2] = d;
            stat_partial[((int64_t)blockIdx.x * ncomp + a) * 2 + 1] = s2;
        }
    }
}

}  // namespace nmfx
