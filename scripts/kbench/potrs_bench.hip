// Timing harness for potrs_panel_kernel (chol.hpp): synthetic packed factor Tm (well-conditioned upper factor U, exact block inverses are not
// needed for timing: a diagonally dominant Tm keeps the values finite), K x N right-hand side.  Correctness is covered by tests/test_gpu_utils.py
// and the ProjectedALS tests through the library.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I nmf.jl_amd/csrc scripts/kbench/potrs_bench.hip -o scripts/kbench/potrs_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
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
int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 6;
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
