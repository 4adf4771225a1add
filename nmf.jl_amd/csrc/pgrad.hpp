// pgrad.hpp -- device kernels of the projected-gradient sub-solvers of ALSPGrad
// (_alspgrad_updateh!, src/alspgrad.jl:86-191; _alspgrad_updatew!, :242-347).
// The back-tracking state machine (alpha, decr_alpha, it, break/continue) lives on the device
// so a batch of back-tracking steps can be enqueued without a host round trip per step.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace nmfx {

struct PgState {
    double alpha;        // step size, always a value of T
    double red[4];       // reduced scalars of the current step: <G,D>, <Gram D,D>, ||Zp-Zn||^2, projgradnorm^2
    int decr_alpha;
    int it;              // back-tracking steps done in the current inner iteration
    int idle;            // 1: no back-tracking in progress (kernels of further steps are no-ops)
    int converged;       // projgradnorm < tolg at the current inner iteration
    int action;          // what pg_apply must do for step `action_step`: 1 Z<-Zn, 2 Z<-Zp, 3 Zp<-Zn
    int action_step;
    int nonfinite;       // alpha became non-finite ("alpha is not finite", src/alspgrad.jl:140,296)
    int zp_valid;        // Zp holds a previous trial point (false at it == 1 where Hp == H)
    long long backtracks;
};

// partial[blk] = sum over this block of g^2 where (g < 0 or z > 0)   (projgradnorm, src/alspgrad.jl:9-19)
template <typename T>
__global__ void pg_norm_kernel(const T *G, const T *Z, int64_t count, double *partial) {
    __shared__ double sm[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const T g = G[i];
        if (g < (T)0 || Z[i] > (T)0) s += (double)(T)(g * g);
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

// red[slot] = sum of n partials (fixed order).  One block.
__global__ void pg_sum_kernel(const double *partial, int n, int nslots, double *red, int slot0, const int *idle) {
    if (idle != nullptr && *reinterpret_cast<const volatile int *>(idle) != 0) return;
    __shared__ double sm[4];
    for (int sl = 0; sl < nslots; ++sl) {
        double v = 0.0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) v += partial[(int64_t)sl * n + i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int q = 0; q < (int)(blockDim.x >> 6); ++q) t += sm[q];
            red[slot0 + sl] = t;
        }
    }
}

// start of an inner iteration (src/alspgrad.jl:129-137): converged if projgradnorm < tolg, else arm back-tracking
template <typename T> __global__ void pg_begin_kernel(PgState *st, T tolg) {
    const T pgnrm = sqrt((T)st->red[3]);
    const bool conv = pgnrm < tolg;
    st->converged = conv ? 1 : 0;
    st->idle = conv ? 1 : 0;
    st->it = 0;
    st->zp_valid = 0;
    st->action = 0;
}

// Zn = max(Z - alpha*G, 0); D = Zn - Z; partial[blk] = <G, D>   (src/alspgrad.jl:142-150)
template <typename T>
__global__ void pg_project_kernel(const T *Z, const T *G, T *Zn, T *D, int64_t count, PgState *st, double *partial) {
    if (*reinterpret_cast<volatile int *>(&st->idle) != 0) return;
    __shared__ double sm[4];
    const T alpha = (T)st->alpha;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const T z = Z[i], g = G[i];
        T v = z - alpha * g;
        v = (v > (T)0) ? v : ((v != v) ? v : (T)0);
        const T d = v - z;
        Zn[i] = v;
        D[i] = d;
        s += (double)(T)(g * d);
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

// partial[blk] = <GD, D>; partial[nblk + blk] = ||Zprev - Zn||^2 with Zprev = Zp if a previous trial exists else Z
template <typename T>
__global__ void pg_dots_kernel(const T *GD, const T *D, const T *Z, const T *Zp, const T *Zn, int64_t count, PgState *st,
                               double *partial) {
    if (*reinterpret_cast<volatile int *>(&st->idle) != 0) return;
    __shared__ double sm[8];
    const T *prev = st->zp_valid ? Zp : Z;
    double s = 0.0, q2 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const T d = D[i];
        s += (double)(T)(GD[i] * d);
        const T e = prev[i] - Zn[i];
        q2 += (double)(T)(e * e);
    }
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off, 64); q2 += __shfl_down(q2, off, 64); }
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = s; sm[4 + (threadIdx.x >> 6)] = q2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int q = 0; q < (int)(blockDim.x >> 6); ++q) { a += sm[q]; b += sm[4 + q]; }
        partial[blockIdx.x] = a;
        partial[gridDim.x + blockIdx.x] = b;
    }
}

// The branch logic of one back-tracking step (src/alspgrad.jl:155-177).
template <typename T>
__global__ void pg_decide_kernel(PgState *st, T beta, T sigma, T epsT, int traceiter, int step_id) {
    if (st->idle) return;
    T alpha = (T)st->alpha;
    if (!isfinite(alpha)) { st->nonfinite = 1; st->idle = 1; return; }   // :140 (checked at the top of each step)
    const T dv1 = (T)st->red[0], dv2 = (T)st->red[1];
    const bool suff_decr = (((T)1 - sigma) * dv1 + (T)0.5 * dv2) < (T)0;
    st->it += 1;
    st->backtracks += 1;
    int action = 0;
    bool brk = false;
    if (st->it == 1) st->decr_alpha = suff_decr ? 0 : 1;                  // :157-160 (Hp <- H is implicit: zp_valid = 0)
    if (st->decr_alpha) {
        if (suff_decr) { action = 1; brk = true; }                        // :163-165 H <- Hn
        else alpha = alpha * beta;                                        // :167
    } else {
        const T nrm = sqrt((T)st->red[2]);                                // isapprox(Hp, Hn, atol=eps(T)) <=> ||Hp-Hn|| <= eps
        if (!suff_decr || nrm <= epsT) { action = st->zp_valid ? 2 : 0; brk = true; }   // :170-172 H <- Hp
        else { alpha = alpha / beta; action = 3; st->zp_valid = 1; }      // :174-175 Hp <- Hn
    }
    st->alpha = (double)alpha;
    st->action = action;
    st->action_step = step_id;
    if (brk || st->it >= traceiter) st->idle = 1;                         // loop exhausted: Z unchanged (quirk i)
}

template <typename T>
__global__ void pg_apply_kernel(T *Z, T *Zp, const T *Zn, int64_t count, const PgState *st, int step_id) {
    if (st->action_step != step_id) return;
    const int action = st->action;
    if (action == 0) return;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        if (action == 1) Z[i] = Zn[i];
        else if (action == 2) Z[i] = Zp[i];
        else Zp[i] = Zn[i];
    }
}

}  // namespace nmfx
