// Timing harness for potrs_panel_kernel (chol.hpp): synthetic packed factor Tm (well-conditioned upper factor U, exact block inverses are not
// needed for timing: a diagonally dominant Tm keeps the values finite), K x N right-hand side.  Correctness is covered by tests/test_gpu_utils.py
// and the ProjectedALS tests through the library.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I nmf.jl_amd/csrc scripts/kbench/potrs_bench.hip -o scripts/kbench/potrs_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>
#include "gemm_mfma.hpp"
#include "kernels.hpp"
#include "chol.hpp"
using namespace nmfx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <typename T, int NB, int NT> void run(int K, int64_t N, int reps, int force_single = 0) {
    T *Tm, *B, *X, *old; double *part;
    CK(hipMalloc(&Tm, (size_t)K * K * sizeof(T))); CK(hipMalloc(&B, (size_t)K * N * sizeof(T))); CK(hipMalloc(&X, (size_t)K * N * sizeof(T)));
    CK(hipMalloc(&old, (size_t)K * N * sizeof(T))); CK(hipMalloc(&part, (size_t)(N / NB) * K * 2 * sizeof(double)));
    std::vector<T> h((size_t)K * K);
    for (int c = 0; c < K; ++c) for (int r = 0; r < K; ++r) h[r + (size_t)c * K] = (T)((r == c) ? 0.5 : 0.001 * ((r * 7 + c * 3) % 11 - 5));
    CK(hipMemcpy(Tm, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    std::vector<T> hb((size_t)K * N);
    for (auto &v : hb) v = (T)(rand() / (double)RAND_MAX);
    CK(hipMemcpy(B, hb.data(), hb.size() * sizeof(T), hipMemcpyHostToDevice)); CK(hipMemcpy(old, hb.data(), hb.size() * sizeof(T), hipMemcpyHostToDevice));
    const size_t s_bytes = (size_t)K * (NB + 1) * sizeof(T), tp_bytes = (size_t)32 * K * sizeof(T);
    const bool dbuf = !force_single && s_bytes + 2 * tp_bytes <= (size_t)160 * 1024;
    const size_t lds = s_bytes + (dbuf ? 2 : 1) * tp_bytes;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&potrs_panel_kernel<T, NB, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((potrs_panel_kernel<T, NB, NT>), dim3((unsigned)(N / NB)), dim3(NT), lds, 0, Tm, K, B, 1, (int64_t)0, K, X, old, K, 1, dbuf ? 1 : 0, part, K, (const int *)nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (i > 0) best = std::min(best, ms);
    }
    CK(hipGetLastError());
    printf("potrs %s K=%d N=%lld NB=%d threads=%d dbuf=%d lds=%zu: %.1f us (%.1f TF/s of the 2 K^2 N substitution flops)\n", sizeof(T) == 4 ? "f32" : "f64", K, (long long)N, NB, NT, (int)dbuf, lds,
           best * 1e3, 2.0 * K * K * N / (best * 1e-3) / 1e12);
    CK(hipFree(Tm)); CK(hipFree(B)); CK(hipFree(X)); CK(hipFree(old)); CK(hipFree(part));
}
// potrs_strip_kernel: a real SPD problem (A = M M' / K + I), factor and diagonal-block inverses on the host in long double, result checked
// against the host's substitution; then timing.
template <typename T, int NBLK> void run_strip(int64_t N, int reps, int with_stats) {
    const int K = 32 * NBLK, nb = NBLK;
    std::vector<long double> A((size_t)K * K), U((size_t)K * K, 0.0L), D((size_t)K * K, 0.0L);
    std::vector<double> M((size_t)K * K);
    for (auto &v : M) v = rand() / (double)RAND_MAX;
    for (int c = 0; c < K; ++c) for (int r = 0; r < K; ++r) { long double s = 0; for (int l = 0; l < K; ++l) s += M[r + (size_t)l * K] * M[c + (size_t)l * K]; A[r + (size_t)c * K] = s / K + (r == c ? 1.0L : 0.0L); }
    for (int j = 0; j < K; ++j) {                       // A = U'U, U upper
        for (int i = 0; i <= j; ++i) {
            long double s = A[i + (size_t)j * K];
            for (int l = 0; l < i; ++l) s -= U[l + (size_t)i * K] * U[l + (size_t)j * K];
            U[i + (size_t)j * K] = (i == j) ? sqrtl(s) : s / U[i + (size_t)i * K];
        }
    }
    for (int b = 0; b < nb; ++b)                        // D = inv(U_bb), upper
        for (int c = 0; c < 32; ++c) {
            std::vector<long double> x(32, 0.0L);
            for (int r = c; r >= 0; --r) {
                long double s = (r == c) ? 1.0L : 0.0L;
                for (int l = r + 1; l <= c; ++l) s -= U[32 * b + r + (size_t)(32 * b + l) * K] * x[l];
                x[r] = s / U[32 * b + r + (size_t)(32 * b + r) * K];
            }
            for (int r = 0; r <= c; ++r) D[32 * b + r + (size_t)(32 * b + c) * K] = x[r];
        }
    std::vector<T> hU((size_t)K * K), hD((size_t)K * K), hb((size_t)K * N), ho((size_t)K * N);
    for (size_t e = 0; e < hU.size(); ++e) { hU[e] = (T)U[e]; hD[e] = (T)D[e]; }
    for (auto &v : hb) v = (T)(rand() / (double)RAND_MAX - 0.3);
    for (auto &v : ho) v = (T)(rand() / (double)RAND_MAX);
    T *dU, *dD, *Tp, *B, *X, *old; double *part;
    CK(hipMalloc(&dU, hU.size() * sizeof(T))); CK(hipMalloc(&dD, hD.size() * sizeof(T))); CK(hipMalloc(&Tp, (size_t)strip_pack_elems(nb) * sizeof(T)));
    CK(hipMalloc(&B, hb.size() * sizeof(T))); CK(hipMalloc(&X, hb.size() * sizeof(T))); CK(hipMalloc(&old, hb.size() * sizeof(T)));
    CK(hipMalloc(&part, (size_t)(N / STRIP_COLS) * K * 2 * sizeof(double)));
    CK(hipMemcpy(dU, hU.data(), hU.size() * sizeof(T), hipMemcpyHostToDevice)); CK(hipMemcpy(dD, hD.data(), hD.size() * sizeof(T), hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hb.data(), hb.size() * sizeof(T), hipMemcpyHostToDevice)); CK(hipMemcpy(old, ho.data(), ho.size() * sizeof(T), hipMemcpyHostToDevice));
    hipLaunchKernelGGL((potrs_strip_pack_kernel<T>), dim3(256), dim3(256), 0, 0, dU, dD, Tp, (int64_t)K, K, nb, (const int *)nullptr);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((potrs_strip_kernel<T, NBLK>), dim3((unsigned)(N / STRIP_COLS)), dim3(64 * STRIP_WAVES), 0, 0, Tp, B, 1, (int64_t)0, (int64_t)K, X, with_stats ? old : (const T *)nullptr, 1,
                           with_stats ? part : (double *)nullptr, K, (const int *)nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (i > 0) best = std::min(best, ms);
    }
    CK(hipGetLastError());
    std::vector<T> hx(hb.size());
    CK(hipMemcpy(hx.data(), X, hx.size() * sizeof(T), hipMemcpyDeviceToHost));
    std::vector<double> hp((size_t)(N / STRIP_COLS) * K * 2);
    CK(hipMemcpy(hp.data(), part, hp.size() * sizeof(double), hipMemcpyDeviceToHost));
    // host check on the first 128 and the last 64 columns
    double worst = 0.0, worst_stat = 0.0;
    std::vector<int64_t> cols;
    for (int64_t c = 0; c < 128 && c < N; ++c) cols.push_back(c);
    for (int64_t c = N - 64; c < N; ++c) cols.push_back(c);
    std::vector<double> ref((size_t)K);
    std::vector<double> sd((size_t)K, 0.0), ss((size_t)K, 0.0);
    for (int64_t c : cols) {
        std::vector<long double> y(K);
        for (int r = 0; r < K; ++r) { long double s = hb[r + (size_t)c * K]; for (int l = 0; l < r; ++l) s -= U[l + (size_t)r * K] * y[l]; y[r] = s / U[r + (size_t)r * K]; }
        for (int r = K - 1; r >= 0; --r) { long double s = y[r]; for (int l = r + 1; l < K; ++l) s -= U[r + (size_t)l * K] * y[l]; y[r] = s / U[r + (size_t)r * K]; }
        for (int r = 0; r < K; ++r) {
            const double want = std::max((double)y[r], 0.0), got = (double)hx[r + (size_t)c * K];
            worst = std::max(worst, std::fabs(want - got) / (1.0 + std::fabs(want)));
            if (c < 64) { const T d = hx[r + (size_t)c * K] - ho[r + (size_t)c * K], sp = hx[r + (size_t)c * K] + ho[r + (size_t)c * K]; sd[r] += (double)(T)(d * d); ss[r] += (double)(T)(sp * sp); }
        }
    }
    if (with_stats) for (int r = 0; r < K; ++r) {
        worst_stat = std::max(worst_stat, std::fabs(hp[(size_t)r * 2] - sd[r]) / (1e-30 + sd[r]));
        worst_stat = std::max(worst_stat, std::fabs(hp[(size_t)r * 2 + 1] - ss[r]) / (1e-30 + ss[r]));
    }
    printf("potrs_strip %s K=%d N=%lld stats=%d: %.1f us (%.1f TF/s of 2 K^2 N); max rel err vs host substitution %.2e, statistics panel 0 %.2e\n", sizeof(T) == 4 ? "f32" : "f64", K,
           (long long)N, with_stats, best * 1e3, 2.0 * K * K * N / (best * 1e-3) / 1e12, worst, worst_stat);
    CK(hipFree(dU)); CK(hipFree(dD)); CK(hipFree(Tp)); CK(hipFree(B)); CK(hipFree(X)); CK(hipFree(old)); CK(hipFree(part));
}
int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 6;
    if (argc > 2 && atoi(argv[2]) == 1) {
        run_strip<float, 8>(16384, reps, 0); run_strip<float, 8>(16384, reps, 1); run_strip<float, 8>(131072, reps, 1);
        run_strip<float, 4>(16384, reps, 1); run_strip<float, 2>(4096, reps, 1); run_strip<float, 6>(8192, reps, 1);
        run_strip<double, 8>(8192, reps, 1); run_strip<double, 4>(8192, reps, 1); run_strip<double, 2>(8192, reps, 0);
        run_strip<float, 10>(8192, reps, 1); run_strip<float, 12>(16384, reps, 1); run_strip<float, 14>(8192, reps, 1); run_strip<float, 16>(16384, reps, 1);
        return 0;
    }
    run<float, 64, 512>(256, 16384, reps);
    run<float, 32, 256>(256, 16384, reps, 1);
    run<float, 32, 512>(256, 16384, reps, 1);
    run<float, 32, 256>(256, 16384, reps, 0);
    run<float, 64, 512>(128, 16384, reps);
    run<float, 32, 256>(128, 16384, reps, 1);
    run<float, 32, 256>(128, 16384, reps, 0);
    run<float, 64, 512>(64, 4096, reps);
    run<float, 32, 256>(64, 4096, reps);
    run<double, 32, 512>(128, 8192, reps);
    run<double, 16, 256>(128, 8192, reps);
    run<double, 32, 512>(256, 8192, reps);
    run<double, 16, 256>(256, 8192, reps, 1);
    return 0;
}
