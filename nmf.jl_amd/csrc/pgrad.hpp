// pgrad.hpp -- device side of the projected-gradient sub-solvers of ALSPGrad
// (_alspgrad_updateh!, src/alspgrad.jl:86-191; _alspgrad_updatew!, :242-347).
//
// None of the reference's work arrays Hn, Hp, D, WtWD (src/alspgrad.jl:41-60) exists on the device: each is a pure
// function of (Z, G, alpha),
//     Zn(alpha) = max(Z - alpha*G, 0),   D(alpha) = Zn(alpha) - Z,   Zp = Zn(alpha_prev),
// so a back-tracking step is ONE MFMA GEMM launch -- Gram*D(alpha) with D computed inside the operand loader, and the
// three scalars <G,D>, <Gram D,D>, ||Zp-Zn||^2 reduced in its epilogue -- plus a one-block decision kernel that carries
// the state machine (alpha, decr_alpha, it, accept / restore / grow, "20 steps exhausted => unchanged", non-finite alpha).
// A batch of steps is enqueued without a host round trip; steps after the loop breaks are no-ops (PgState::idle).
//
// Round 6: the iterate lives in THREE rotating buffer sets (Z, G)[0..2].  A step reads the current point from set `base` and its
// epilogue -- which holds z, g and Gram*D of every element anyway -- writes the TRIAL point Zn(alpha) and ITS gradient G + Gram*D
// (G(Z + D) = G(Z) + Gram*D) into one of the two other sets, together with that point's projgradnorm^2 as a fourth sum.  Accepting a
// trial point is then the decision kernel switching `base` (and taking the norm from the step's sums): the element-wise pass that
// applied the accepted step, advanced the gradient and reduced the norm between two inner iterations (round 5's pg_advance_kernel:
// five arrays, 10.5 % of a C5 outer iteration) no longer exists -- one launch less per inner iteration.  What it buys, measured
// (profiles/r06_alspgrad_*): NOTHING on the 1-GPU C5 shard (211.4 against 211.9 ms per outer iteration: the second 134 MB store of a
// W-side step and the fourth sum cost its un-overlapped epilogue what the pass cost, +14 us on the average step), 71.0 against 72.2 ms
// at the 8-rank shard shape, where the passes were launch-bound.  The middle form -- only G in rotating sets (G + Gram*D stored where
// Gram*D was), Z updated in place by a three-array pass -- was measured too: 212.9 / 75.7 ms, no better.  The trial step's epilogue,
// not the pass, is what a C5 iteration waits for; the sets stay because they remove a kernel and a launch per inner iteration.
// Same arithmetic per element (Zn = max(Z - alpha G, 0) with the accepted alpha, G + Gram*D: one add in T): identical Z, G bits.
#pragma once
#include <hip/hip_runtime.h>

#include "gemm_mfma.hpp"
#include "kernels.hpp"
#include "peer.hpp"

namespace nmfx {

struct PgState {
    double alpha;        // current step size, always a value of T                         (alspgrad.jl:119)
    double alpha_prev;   // step size of the previous trial point Zp (valid if zp_valid)
    double alpha_apply;  // step size whose Zn must become Z (set with apply = 1)
    double red[4];       // <G,D>, <Gram D,D>, ||Zp-Zn||^2, projgradnorm^2
    int decr_alpha;
    int it;              // back-tracking steps done in the current inner iteration
    int idle;            // 1: no back-tracking in progress (kernels of further steps are no-ops)
    int converged;       // projgradnorm < tolg at the current inner iteration
    int apply;           // 1: Z <- max(Z - alpha_apply*G, 0) pending (pg_apply_kernel clears it)
    int nonfinite;       // alpha became non-finite ("alpha is not finite", alspgrad.jl:140,296)
    int zp_valid;        // a previous trial point exists (false at it == 1 where Hp == H)
    int gate;            // != 0: inner iterations enqueued ahead of the host's knowledge are no-ops (= converged | halt)
    long long backtracks;
    int halt;            // the line search of the current inner iteration needs more steps than were enqueued: the host takes over
    int t_inner;         // executed inner iterations of this sub-solve (the converged one included, like the reference's t)
    int gd_sel;          // (round 5: which of the two Gram*D buffers belonged to the accepted trial point; unused)
    int hist[8];         // finished line searches by number of steps (last bin: 8 or more); NMFX_PG_HIST=1 prints it per sub-solve
    int base;            // buffer set (0..2) that holds the current point Z and its gradient G; step `it` of a search writes its trial
                         // point into set (base + 1 + (it & 1)) % 3
    int pad0;
    double pgn[2];       // projgradnorm^2 of the trial points in the two other sets (by it & 1)
};

template <typename T> __device__ __forceinline__ T pg_trial(T z, T g, T alpha) {
    const T v = z - alpha * g;
    return (v > (T)0) ? v : ((v != v) ? v : (T)0);          // max(v, 0), NaN propagates
}

// G = acc - B (src/alspgrad.jl:124-127, 280-283) + per-block partial of projgradnorm^2 (alspgrad.jl:9-19:
// sum of g^2 where g < 0 or z > 0; term in T, sum in Float64).
template <typename T> struct EpiGradNorm {
    const T *B;
    const T *Z;
    T *G;
    int64_t ld;
    double *partial;   // one per block
    double sum;
    const PgState *st = nullptr;   // != nullptr: Z and G are set 0 of the rotating sets, the current point is set st->base
    int64_t set_stride = 0;
    struct Pre { T b, z; };
    static constexpr bool EARLY = true, HEAVY = true;
    rsrc_t rb, rz, rg;
    LaneAddr<T> la;
    __device__ __forceinline__ void setup(int, const TileCtx &t) {
        const int64_t off = (st != nullptr) ? (int64_t)st->base * set_stride : 0;
        rb = tile_rsrc(B, ld, t); rz = tile_rsrc(Z + off, ld, t); rg = tile_rsrc(G + off, ld, t);
        la.init(t, ld);
    }
    __device__ __forceinline__ void begin() { sum = 0.0; }
    __device__ __forceinline__ Pre prefetch(int ro, int co) const {
        const uint32_t so = la.soff(ro, co);
        return Pre{buf_ld<T>(rb, la.lb, so), buf_ld<T>(rz, la.lb, so)};
    }
    __device__ __forceinline__ void apply(int ro, int co, T v, int, const Pre &pre) {
        const T g = v - pre.b;
        buf_st(rg, la.lb, la.soff(ro, co), g);
        if (g < (T)0 || pre.z > (T)0) sum += (double)(T)(g * g);
    }
    template <int MT, int TC, int WGR, int WGC> __device__ __forceinline__ void finish(double *smem, const TileCtx &t) {
        block_sum_store(sum, smem, t.tid, t.nthreads, partial + t.bid);
    }
};

// One back-tracking step: acc = (Gram * D(alpha))(r, c).  Per block: partial[4*bid + {0,1,2,3}] =
//   <G, D>, <Gram D, D>, ||Zprev - Zn||^2   with Zprev = Zn(alpha_prev) if a previous trial exists, else Z
// (src/alspgrad.jl:150-152 and the isapprox of :170), and projgradnorm^2 of the TRIAL point (alspgrad.jl:9-19 on Zn and G + Gram*D).
// Z, G of the current point are read from set st->base; the trial point Zn and its gradient G + Gram*D (G(Z + D) = G(Z) + Gram*D) go
// to set (base + 1 + (it & 1)) % 3 (it = steps done so far in this search: the accepted trial point is this step's or the previous
// one's), so accepting is a switch of `base` in the decision kernel -- no pass over Z and G.
template <typename T> struct EpiPgStep {
    T *ZS;             // set 0 of the Z sets (same layout and leading dimension in every set)
    T *GS;             // set 0 of the G sets
    int64_t set_stride;
    int64_t ld;
    const PgState *st;
    double *partial;
    T alpha, alpha_prev;
    int zp_valid;
    double s1, s2, s3, s4;
    struct Pre { T z, g; };
    static constexpr bool EARLY = true, HEAVY = true;
    rsrc_t rz, rg, rzn, rgn;
    LaneAddr<T> la;
    __device__ __forceinline__ void setup(int, const TileCtx &t) {
        const int b = st->base;
        int slot = b + 1 + (st->it & 1);
        slot = (slot >= 3) ? slot - 3 : slot;
        rz = tile_rsrc(ZS + (int64_t)b * set_stride, ld, t); rg = tile_rsrc(GS + (int64_t)b * set_stride, ld, t);
        rzn = tile_rsrc(ZS + (int64_t)slot * set_stride, ld, t); rgn = tile_rsrc(GS + (int64_t)slot * set_stride, ld, t);
        la.init(t, ld);
    }
    __device__ __forceinline__ void begin() {
        alpha = (T)st->alpha;
        alpha_prev = (T)st->alpha_prev;
        zp_valid = st->zp_valid;
        s1 = s2 = s3 = s4 = 0.0;
    }
    __device__ __forceinline__ Pre prefetch(int ro, int co) const {
        const uint32_t so = la.soff(ro, co);
        return Pre{buf_ld<T>(rz, la.lb, so), buf_ld<T>(rg, la.lb, so)};
    }
    __device__ __forceinline__ void apply(int ro, int co, T v, int, const Pre &pre) {
        const T zn = pg_trial(pre.z, pre.g, alpha);
        const T gn = pre.g + v;
        const uint32_t so = la.soff(ro, co);
        buf_st(rzn, la.lb, so, zn);
        buf_st(rgn, la.lb, so, gn);
        const T d = zn - pre.z;
        const T zprev = zp_valid ? pg_trial(pre.z, pre.g, alpha_prev) : pre.z;
        const T e = zprev - zn;
        s1 += (double)(T)(pre.g * d);
        s2 += (double)(T)(v * d);
        s3 += (double)(T)(e * e);
        if (gn < (T)0 || zn > (T)0) s4 += (double)(T)(gn * gn);
    }
    template <int MT, int TC, int WGR, int WGC> __device__ __forceinline__ void finish(double *smem, const TileCtx &t) {
        block_sum_store(s1, smem, t.tid, t.nthreads, partial + 4 * t.bid);
        block_sum_store(s2, smem, t.tid, t.nthreads, partial + 4 * t.bid + 1);
        block_sum_store(s3, smem, t.tid, t.nthreads, partial + 4 * t.bid + 2);
        block_sum_store(s4, smem, t.tid, t.nthreads, partial + 4 * t.bid + 3);
    }
};

// fixed-order sum of `n` partials with stride `stride` starting at `off` (one block of 256 threads)
__device__ __forceinline__ double pg_block_sum(const double *partial, int n, int stride, int off, double *sm) {
    double v = 0.0;
    // (eight loads in flight, then the adds in index order: a plain `v += partial[..]` loop over a run-time count waits for every load
    // before it issues the next -- n / 256 dependent round trips in a kernel that is nothing but this sum)
    const int step = (int)blockDim.x;
    for (int i0 = threadIdx.x; i0 < n; i0 += 8 * step) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = (i0 + u * step < n) ? i0 + u * step : i0;
            x[u] = partial[(int64_t)i * stride + off];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + u * step < n) v += x[u];
    }
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int q = 0; q < (int)(blockDim.x >> 6); ++q) t += sm[q];
    return t;
}

// the four sums of a back-tracking step at once: wave w < 4 of the block sums slot w (same per-slot order for every launch)
constexpr int PG_NSUM = 4;
__device__ __forceinline__ void pg_block_sum3(const double *partial, int n, double *out3) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (w < PG_NSUM) {
        double v = 0.0;
        // (the W-side trial step of C5 leaves 2048 block partials: 32 per lane -- as a chain of dependent loads that was most of a
        // decision kernel's time; eight in flight, added in index order: the same sum)
        for (int i0 = lane; i0 < n; i0 += 8 * 64) {
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = (i0 + 64 * u < n) ? i0 + 64 * u : i0;
                x[u] = partial[(int64_t)i * PG_NSUM + w];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + 64 * u < n) v += x[u];
        }
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) out3[w] = v;
    }
    __syncthreads();
}

// out[w] = sum of partial[i * nslot + w] over the launch's n blocks (nslot = 1 or 3), in the order pg_begin_kernel / pg_decide_kernel
// use themselves: the rank-LOCAL sums of a sharded sub-solve on a transport without the in-kernel all-reduce (RCCL, the in-process
// group).  The collective that follows then moves nslot doubles whatever the ranks' block counts are -- all-reducing the per-block
// partials themselves (round 3) used a rank-dependent count as soon as ragged column shards straddle a 256-column boundary.
static __global__ void pg_local_sum_kernel(const double *partial, int n, int nslot, double *out) {
    __shared__ double sm[4];
    if (nslot == PG_NSUM) { pg_block_sum3(partial, n, out); return; }
    const double s = pg_block_sum(partial, n, 1, 0, sm);
    if (threadIdx.x == 0) out[0] = s;
}

// start of an inner iteration (src/alspgrad.jl:129-137): red[3] = projgradnorm^2 from the gradient GEMM's partials
// (sharded: the rank's own sum first, then the ranks' sums in rank order -- inside this kernel on the peer transport (`tiny`),
// by an all-reduce of ONE double in front of it otherwise); converged if < tolg, else arm back-tracking.
template <typename T> __global__ void pg_begin_kernel(PgState *st, const double *partial, int n_local, T tolg, TinyAR tiny) {
    if (st->gate) return;
    __shared__ double sm[4];
    __shared__ double tsm[PEER_TINY_MAX];
    if (n_local > 0) {
        double s = pg_block_sum(partial, n_local, 1, 0, sm);
        if (tiny.n > 1) tiny_allreduce(tiny, &s, 1, tsm);
        if (threadIdx.x == 0) st->red[3] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const T pgnrm = sqrt((T)st->red[3]);
        const bool conv = pgnrm < tolg;
        st->converged = conv ? 1 : 0;
        st->idle = conv ? 1 : 0;
        st->it = 0;
        st->zp_valid = 0;
        st->apply = 0;
        st->t_inner += 1;
        if (conv) st->gate = 1;
    }
}

// the host finished a halted line search: speculation may continue
static __global__ void pg_resume_kernel(PgState *st) {
    st->halt = 0;
    st->gate = st->converged ? 1 : 0;
}

// The branch logic of one back-tracking step (src/alspgrad.jl:155-177); n_local > 0: sum the step's partials first.
template <typename T>
__global__ void pg_decide_kernel(PgState *st, const double *partial, int n_local, T beta, T sigma, T epsT, int traceiter, int last_enqueued, TinyAR tiny) {
    if (st->idle) return;
    __shared__ double r3[PG_NSUM];
    __shared__ double tsm[PEER_TINY_MAX];
    if (n_local > 0) pg_block_sum3(partial, n_local, r3);
    double v3[PG_NSUM] = {r3[0], r3[1], r3[2], r3[3]};
    if (n_local > 0 && tiny.n > 1) tiny_allreduce(tiny, v3, PG_NSUM, tsm);   // the ranks' sums, added in rank order (every rank decides on the same bits)
    if (threadIdx.x != 0) return;
    if (n_local > 0) { st->red[0] = v3[0]; st->red[1] = v3[1]; st->red[2] = v3[2]; st->pgn[st->it & 1] = v3[3]; }
    T alpha = (T)st->alpha;
    if (!isfinite(alpha)) { st->nonfinite = 1; st->idle = 1; st->gate = 1; return; }   // :140 (the step's sums are garbage then)
    const T dv1 = (T)st->red[0], dv2 = (T)st->red[1];
    const bool suff_decr = (((T)1 - sigma) * dv1 + (T)0.5 * dv2) < (T)0;
    const int slot = st->it & 1;                                          // which of the two other sets this step's trial point went to
    int accept = -1;
    st->it += 1;
    st->backtracks += 1;
    bool brk = false;
    if (st->it == 1) st->decr_alpha = suff_decr ? 0 : 1;                  // :157-160 (Hp <- H is implicit: zp_valid = 0)
    if (st->decr_alpha) {
        if (suff_decr) { accept = slot; brk = true; }                     // :163-165 H <- Hn
        else alpha = alpha * beta;                                        // :167
    } else {
        const T nrm = sqrt((T)st->red[2]);                                // isapprox(Hp, Hn, atol=eps(T)) <=> ||Hp-Hn|| <= eps
        if (!suff_decr || nrm <= epsT) {                                  // :170-172 H <- Hp
            if (st->zp_valid) accept = slot ^ 1;
            brk = true;
        } else {                                                          // :174-175 alpha /= beta; Hp <- Hn
            st->alpha_prev = (double)alpha;
            st->zp_valid = 1;
            alpha = alpha / beta;
        }
    }
    st->alpha = (double)alpha;
    if (accept >= 0) {                                                    // the accepted trial point becomes the current one: its set, its norm
        int nb = st->base + 1 + accept;
        st->base = (nb >= 3) ? nb - 3 : nb;
        st->red[3] = st->pgn[accept];
    }
    if (brk || st->it >= traceiter) { st->idle = 1; st->hist[st->it < 8 ? st->it - 1 : 7] += 1; }   // loop exhausted: Z unchanged (quirk i)
    // last step enqueued for this inner iteration and the search is still running: it hands over to the host (every inner
    // iteration enqueued behind this one becomes a no-op until the host has finished the search)
    else if (last_enqueued) { st->halt = 1; st->gate = 1; }
}

// Z (rows x cols, leading dimension ld) <-> set `st->base` (out) / set 0 (in) of the rotating Z sets, same layout: the copy-in at the start
// of a sub-solve and the copy-out of the accepted point at its end (2 passes per sub-solve; the per-iteration passes are gone).
// grid = (row chunks, column groups) like the element-wise passes it replaces: a block walks a contiguous row range of its columns.
template <typename T> __global__ void pg_copy_kernel(T *Z, T *ZS, int64_t set_stride, int64_t rows, int64_t cols, int64_t ld, const PgState *st, int out) {
    T *S = ZS + (out ? (int64_t)st->base * set_stride : 0);
    const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = (r0 + per < rows) ? r0 + per : rows;
    for (int64_t c = blockIdx.y; c < cols; c += gridDim.y) {
        T *z = Z + c * ld, *q = S + c * ld;
        if (out) { for (int64_t r = r0 + threadIdx.x; r < r1; r += blockDim.x) z[r] = q[r]; }
        else { for (int64_t r = r0 + threadIdx.x; r < r1; r += blockDim.x) q[r] = z[r]; }
    }
}

}  // namespace nmfx
