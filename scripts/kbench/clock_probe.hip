// Measures the shader clock while the big GEMM runs: a 1-wave sampler kernel on a second stream reads
// clock64() (s_memtime, shader cycles) and wall_clock64() (s_memrealtime, 100 MHz) every few microseconds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_mfma.hpp"
using namespace nmfx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void sampler(long long *out, int n, int spin) {
    for (int i = 0; i < n; ++i) {
        out[2 * i] = clock64();
        out[2 * i + 1] = wall_clock64();
        for (int s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(64);
    }
}

template <int SIGNED> void probe(const char *tag) {
    const int64_t R = 16384, C = 256, Kd = 16384; const int splits = 2;
    float *A, *B, *D;
    CK(hipMalloc(&A, (size_t)R * Kd * 4)); CK(hipMalloc(&B, (size_t)C * Kd * 4)); CK(hipMalloc(&D, (size_t)R * C * splits * 4));
    std::vector<float> h((size_t)R * Kd);
    for (auto &v : h) v = (float)(rand() / (double)RAND_MAX) - (SIGNED ? 0.5f : 0.0f);
    CK(hipMemcpy(A, h.data(), (size_t)R * Kd * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data(), (size_t)C * Kd * 4, hipMemcpyHostToDevice));
    GemmArgs<float> g; g.A = A; g.B = B; g.lda = Kd; g.ldb = Kd; g.tiles_r = R / 128; g.tiles_c = C / 128; g.splits = splits;
    g.kchunk = Kd / splits; g.c_fastest = 1; g.done = nullptr;
    EpiStore<float> e{D, C, R * C, nullptr};
    long long *dS; const int NS = 40000; CK(hipMalloc(&dS, NS * 16)); CK(hipMemset(dS, 0, NS * 16));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // idle clock first
    hipLaunchKernelGGL(sampler, dim3(1), dim3(64), 0, s2, dS, 200, 40);
    CK(hipStreamSynchronize(s2));
    std::vector<long long> hs(2 * NS);
    CK(hipMemcpy(hs.data(), dS, NS * 16, hipMemcpyDeviceToHost));
    double idle = (double)(hs[2 * 199] - hs[2 * 10]) / (double)(hs[2 * 199 + 1] - hs[2 * 10 + 1]) * 100.0;
    hipLaunchKernelGGL(sampler, dim3(1), dim3(64), 0, s2, dS, NS, 2);
    const int NG = 60;
    for (int i = 0; i < NG; ++i) {
        if (i == NG - 1) CK(hipEventRecord(e0, s1));
        hipLaunchKernelGGL((gemm_mfma_kernel<float, KCONTIG, KCONTIG, 128, 128, 2, 2, EpiStore<float>>), dim3(512), dim3(256), 0, s1, g, e);
        if (i == NG - 1) CK(hipEventRecord(e1, s1));
    }
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(hs.data(), dS, NS * 16, hipMemcpyDeviceToHost));
    // clock over windows of 100 samples
    printf("%s: idle-chip shader clock %.0f MHz; GEMM %.1f us = %.1f TF/s; clock during GEMM launches (MHz per window):", tag, idle, ms * 1e3,
           2.0 * R * C * Kd / (ms * 1e-3) / 1e12);
    for (int w = 0; w + 2000 < NS; w += 2000) {
        const double mhz = (double)(hs[2 * (w + 2000)] - hs[2 * w]) / (double)(hs[2 * (w + 2000) + 1] - hs[2 * w + 1]) * 100.0;
        printf(" %.0f", mhz);
    }
    printf("  [total sampled %.2f ms]\n", (hs[2 * (NS - 1) + 1] - hs[1]) / 100e3);
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(D)); CK(hipFree(dS));
}
int main() { probe<1>("signed [-0.5,0.5)"); probe<0>("non-negative [0,1)"); return 0; }
