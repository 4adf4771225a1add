// spa_impl.hpp -- spa(X, k), the successive projection algorithm behind nnmf(init = :spa) / nnmf(alg = :spa)
// (src/spa.jl:38-63; src/interf.jl:50-51, 73-77), on the resident X.  Widening beyond SURVEY.md section 8f: the last init / alg
// option of nnmf that did not run on the device.
//
//   R = X ./ sum(X, dims=1)                                     spa.jl:41      one pass (column sums, scaling, column norms)
//   k times:  a = argmax_j ||R[:, j]||^2                        spa.jl:50      one block (first index on ties)
//             p = R[:, a];  R -= p (p'R) ./ (p'p)               spa.jl:53-54   one pass: a block per column does the dot product,
//                                                                              the update and the column's new norm
//   W = X[:, anchors]                                           spa.jl:58
//   H = nonneg_lsq(W, X, alg = :fnnls); projectnn!(H)           spa.jl:61-62   NonNegLeastSquares.jl is not vendored.  fnnls returns
//       THE minimiser of ||X[:, j] - W h||, h >= 0; here it is reached by exact coordinate minimisation on the normal equations
//       (W'W, W'X from the hot path's own launch; the row-parallel sweep of CoordinateDescent, cd.hpp) iterated until no entry
//       of H moves by more than tol * max|H| -- the same fixed point, approached instead of hit (documented deviation; the tests
//       compare with an exact active-set solver).
// Column sums / norms / dot products accumulate in Float64 and are rounded to T where the reference holds a T (its sums run in
// T in Julia's pairwise order, which cannot be restated bit for bit anyway); ties in the argmax go to the first index.
#pragma once
#include "cd_impl.hpp"
#include "solver.hpp"

namespace nmfx {

__device__ __forceinline__ double spa_block_sum(double v, double *sm) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int q = 0; q < (int)(blockDim.x >> 6); ++q) t += sm[q];
    return t;
}

// one block per column j < n: R[:, j] = X[:, j] / sum(X[:, j]);  nrm[j] = ||R[:, j]||^2
template <typename T> __global__ __launch_bounds__(256) void spa_normalize_kernel(T *R, const T *X, int64_t p, int64_t ld, double *nrm) {
    __shared__ double sm[4];
    const int64_t j = blockIdx.x;
    const T *x = X + j * ld;
    T *r = R + j * ld;
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < p; i += blockDim.x) s += (double)x[i];
    const T cs = (T)spa_block_sum(s, sm);
    double q = 0.0;
    for (int64_t i = threadIdx.x; i < p; i += blockDim.x) {
        const T v = x[i] / cs;
        r[i] = v;
        q += (double)(T)(v * v);
    }
    q = spa_block_sum(q, sm);
    if (threadIdx.x == 0) nrm[j] = q;
}

// one block: a = argmax nrm (first index on ties) -> anchors[step];  pvec = R[:, a];  scal[0] = p'p
template <typename T>
__global__ __launch_bounds__(1024) void spa_pick_kernel(const double *nrm, int64_t n, const T *R, int64_t p, int64_t ld, long long *anchors, int step,
                                                        T *pvec, double *scal) {
    __shared__ double sv[16];
    __shared__ long long si[16];
    __shared__ long long best_j;
    double bv = -1.0;
    long long bi = 0x7fffffffffffffffll;
    for (int64_t j = threadIdx.x; j < n; j += blockDim.x) {
        const double v = nrm[j];
        if (v > bv || (v == bv && j < bi) || (v != v && bv == bv)) { bv = v; bi = j; }   // NaN wins like Julia's argmax
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_down(bv, off, 64);
        const long long oi = __shfl_down(bi, off, 64);
        if (ov > bv || (ov == bv && oi < bi) || (ov != ov && bv == bv)) { bv = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi) || (sv[w] != sv[w] && bv == bv)) { bv = sv[w]; bi = si[w]; }
        best_j = bi;
        anchors[step] = bi;
    }
    __syncthreads();
    const T *col = R + best_j * ld;
    double q = 0.0;
    for (int64_t i = threadIdx.x; i < p; i += blockDim.x) {
        const T v = col[i];
        pvec[i] = v;
        q += (double)(T)(v * v);
    }
    __shared__ double sm[16];
    q = spa_block_sum(q, sm);
    if (threadIdx.x == 0) scal[0] = q;
}

// one block per column j < n: R[:, j] -= p (p'R[:, j]) ./ (p'p);  nrm[j] = ||R[:, j]||^2 of the result
template <typename T>
__global__ __launch_bounds__(256) void spa_project_kernel(T *R, const T *pvec, const double *scal, int64_t p, int64_t ld, double *nrm) {
    __shared__ double sm[4];
    const int64_t j = blockIdx.x;
    T *r = R + j * ld;
    double d = 0.0;
    for (int64_t i = threadIdx.x; i < p; i += blockDim.x) d += (double)(T)(pvec[i] * r[i]);
    const T dot = (T)spa_block_sum(d, sm);
    const T ptp = (T)scal[0];
    double q = 0.0;
    for (int64_t i = threadIdx.x; i < p; i += blockDim.x) {
        const T v = r[i] - (T)(pvec[i] * dot) / ptp;
        r[i] = v;
        q += (double)(T)(v * v);
    }
    q = spa_block_sum(q, sm);
    if (threadIdx.x == 0) nrm[j] = q;
}

// W(i, s) = X(i, anchors[s])
template <typename T> __global__ void spa_gather_kernel(T *W, const T *X, const long long *anchors, int64_t p, int64_t ld, int k) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < p * k; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e % p, s = e / p;
        W[i + s * ld] = X[i + anchors[s] * ld];
    }
}

// out[0] = max |a - b|, out[1] = max |a| over the logical k x n block (ld)
template <typename T> __global__ __launch_bounds__(1024) void spa_maxdiff_kernel(const T *a, const T *b, int64_t k, int64_t n, int64_t ld, double *out) {
    __shared__ double s1[16], s2[16];
    double d = 0.0, m = 0.0;
    for (int64_t e = threadIdx.x; e < k * n; e += blockDim.x) {
        const int64_t i = e % k + (e / k) * ld;
        const double x = (double)a[i], y = (double)b[i];
        d = fmax(d, fabs(x - y));
        m = fmax(m, fabs(x));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { d = fmax(d, __shfl_down(d, off, 64)); m = fmax(m, __shfl_down(m, off, 64)); }
    if ((threadIdx.x & 63) == 0) { s1[threadIdx.x >> 6] = d; s2[threadIdx.x >> 6] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { d = fmax(d, s1[w]); m = fmax(m, s2[w]); }
        out[0] = d;
        out[1] = m;
    }
}

template <typename T> void Solver<T>::spa_init(int max_sweeps, double tol, int64_t *anchors_out, int *sweeps_out) {
    if (!have_X) throw StatusError{NMFX_ERR_STATE, "X has not been uploaded (nmfx_set_X)"};
    if (nranks > 1) throw StatusError{NMFX_ERR_UNSUPPORTED, "spa: the anchor search runs on one GPU (attach no communicator)"};
    if (max_sweeps < 1 || !(tol > 0)) throw StatusError{NMFX_ERR_BAD_ARG, "spa: max_sweeps must be >= 1 and tol positive"};
    HIP_TRY(hipSetDevice(device));
    rsvd_ready = 0;
    Q.ensure((size_t)P * N);                      // R
    work[0].ensure((size_t)P);                    // p
    nd_scratch.ensure((size_t)N + 8);             // column norms | p'p | max-diff pair
    flag_ll.ensure((size_t)K);
    T *R = Q.p, *pvec = work[0].p;
    double *nrm = nd_scratch.p, *scal = nd_scratch.p + N;
    hipLaunchKernelGGL(spa_normalize_kernel<T>, dim3((unsigned)n), dim3(256), 0, stream, R, X.p, p, P, nrm);
    for (int s = 0; s < (int)k; ++s) {
        hipLaunchKernelGGL(spa_pick_kernel<T>, dim3(1), dim3(1024), 0, stream, nrm, n, R, p, P, flag_ll.p, s, pvec, scal);
        hipLaunchKernelGGL(spa_project_kernel<T>, dim3((unsigned)n), dim3(256), 0, stream, R, pvec, scal, p, P, nrm);
    }
    HIP_TRY(hipGetLastError());
    // W = X[:, anchors];  H = 0
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipMemsetAsync(W[i].p, 0, W[i].count * sizeof(T), stream));
        HIP_TRY(hipMemsetAsync(H[i].p, 0, H[i].count * sizeof(T), stream));
    }
    wcur = hcur = 0;
    hipLaunchKernelGGL(spa_gather_kernel<T>, dim3(flat_grid(p * k)), dim3(256), 0, stream, W[0].p, X.p, flag_ll.p, p, P, (int)k);
    // non-negative least squares for every column of X: coordinate minimisation on the normal equations
    precision = NMFX_PREC_FP32;
    wt_times(W[0].p, X.p, true, nullptr);
    int sweeps = 0;
    double host2[2] = {0.0, 0.0};
    const int batch = 8;
    while (sweeps < max_sweeps) {
        const int m = std::min(batch, max_sweeps - sweeps);
        for (int b = 0; b < m; ++b) {
            cd_sweep(SampleView<const T>{H[hcur].p, K, 1}, SampleView<T>{H[hcur ^ 1].p, K, 1}, SampleView<const T>{numH_p, K, 1}, gramW_p, n, (T)0,
                     nullptr);
            hcur ^= 1;
        }
        sweeps += m;
        hipLaunchKernelGGL(spa_maxdiff_kernel<T>, dim3(1), dim3(1024), 0, stream, H[hcur].p, H[hcur ^ 1].p, k, n, K, scal + 1);
        HIP_TRY(hipMemcpyAsync(host2, scal + 1, 2 * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (!(host2[0] > tol * host2[1])) break;      // also leaves on NaN
    }
    std::vector<long long> anc((size_t)k);
    HIP_TRY(hipMemcpyAsync(anc.data(), flag_ll.p, (size_t)k * sizeof(long long), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (anchors_out) for (int64_t s = 0; s < k; ++s) anchors_out[s] = (int64_t)anc[(size_t)s];
    if (sweeps_out) *sweeps_out = sweeps;
    have_F = true;
}

}  // namespace nmfx
