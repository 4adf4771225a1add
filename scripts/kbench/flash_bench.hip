// Development harness for csrc/flash_div.hpp: correctness against a naive double-precision kernel on small shapes, then timing
// at the C3 shape (16384 x 16384, k = 256).   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I nmf.jl_amd/csrc -I scripts/kbench scripts/kbench/flash_bench.hip -o scripts/kbench/flash_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "flash_div.hpp"
using namespace nmfx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// QHt(i, a) and WtQ(a, j) by brute force (one thread per output, double accumulation)
__global__ void ref_qht(double *out, const float *W, const float *H, const float *X, int P, int N, int K, float delta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, a = blockIdx.y;
    if (i >= P) return;
    double s = 0.0;
    for (int j = 0; j < N; ++j) {
        float wh = 0.f;
        for (int b = 0; b < K; ++b) wh += W[(size_t)b * P + i] * H[(size_t)j * K + b];
        s += (double)(X[(size_t)j * P + i] / (wh + delta)) * (double)H[(size_t)j * K + a];
    }
    out[(size_t)a * P + i] = s;
}
__global__ void ref_wtq(double *out, const float *W, const float *H, const float *X, int P, int N, int K, float delta) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, a = blockIdx.y;
    if (j >= N) return;
    double s = 0.0;
    for (int i = 0; i < P; ++i) {
        float wh = 0.f;
        for (int b = 0; b < K; ++b) wh += W[(size_t)b * P + i] * H[(size_t)j * K + b];
        s += (double)(X[(size_t)j * P + i] / (wh + delta)) * (double)W[(size_t)a * P + i];
    }
    out[(size_t)j * K + a] = s;
}

template <int K> void run(int P, int N, int reps, bool check) {
    const float delta = 3.4527e-4f;
    float *W, *H, *X, *Xt, *outW, *outH;
    CK(hipMalloc(&W, (size_t)P * K * 4)); CK(hipMalloc(&H, (size_t)K * N * 4)); CK(hipMalloc(&X, (size_t)P * N * 4)); CK(hipMalloc(&Xt, (size_t)P * N * 4));
    const int xbW = P / 64, xbH = N / 64;
    int spW = 1, spH = 1;
    while (xbW * spW < 256 && (N / 64) % (spW * 2) == 0) spW *= 2;
    while (xbH * spH < 256 && (P / 64) % (spH * 2) == 0) spH *= 2;
    CK(hipMalloc(&outW, (size_t)spW * P * K * 4)); CK(hipMalloc(&outH, (size_t)spH * K * N * 4));
    std::vector<float> h((size_t)std::max((size_t)P * N, (size_t)K * std::max(P, N)));
    srand(1);
    for (auto &v : h) v = (float)(rand() / (double)RAND_MAX);
    CK(hipMemcpy(X, h.data(), (size_t)P * N * 4, hipMemcpyHostToDevice));
    for (auto &v : h) v = (float)(rand() / (double)RAND_MAX);
    CK(hipMemcpy(H, h.data(), (size_t)K * N * 4, hipMemcpyHostToDevice));
    for (auto &v : h) v = (float)(rand() / (double)RAND_MAX);
    CK(hipMemcpy(W, h.data(), (size_t)P * K * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(flash::transpose_kernel, dim3(P / 64, N / 64), dim3(256), 0, 0, Xt, X, (int64_t)P, (int64_t)N);
    CK(hipDeviceSynchronize());
    auto kW = flash::flash_div_kernel<K, true, false, false>;
    auto kH = flash::flash_div_kernel<K, false, true, true>;
    const size_t lds = flash::lds_bytes<K>();
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kW), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kH), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    flash::Args aw{W, P, H, K, X, P, outW, P, (int64_t)P * K, xbW, (N / 64) / spW, delta, nullptr};
    flash::Args ah{H, K, W, P, Xt, N, outH, K, (int64_t)K * N, xbH, (P / 64) / spH, delta, nullptr};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int side = 0; side < 2; ++side) {
        float best = 1e9;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0));
            if (side == 0) hipLaunchKernelGGL(kW, dim3(xbW * spW), dim3(256), lds, 0, aw);
            else hipLaunchKernelGGL(kH, dim3(xbH * spH), dim3(256), lds, 0, ah);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r > 0 || reps == 1) best = std::min(best, ms);
        }
        CK(hipGetLastError());
        printf("K=%d P=%d N=%d %s splits=%d blocks=%d: %.1f us  %.1f TF/s (4pnk)\n", K, P, N, side == 0 ? "W side (QHt)" : "H side (WtQ)",
               side == 0 ? spW : spH, side == 0 ? xbW * spW : xbH * spH, best * 1e3, 4.0 * P * N * K / (best * 1e-3) / 1e12);
    }
    if (check) {
        double *rW, *rH;
        CK(hipMalloc(&rW, (size_t)P * K * 8)); CK(hipMalloc(&rH, (size_t)K * N * 8));
        hipLaunchKernelGGL(ref_qht, dim3((P + 63) / 64, K), dim3(64), 0, 0, rW, W, H, X, P, N, K, delta);
        hipLaunchKernelGGL(ref_wtq, dim3((N + 63) / 64, K), dim3(64), 0, 0, rH, W, H, X, P, N, K, delta);
        CK(hipDeviceSynchronize());
        std::vector<double> a((size_t)P * K), b((size_t)K * N);
        std::vector<float> gw((size_t)spW * P * K), gh((size_t)spH * K * N);
        CK(hipMemcpy(a.data(), rW, a.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), rH, b.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(gw.data(), outW, gw.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gh.data(), outH, gh.size() * 4, hipMemcpyDeviceToHost));
        double ew = 0, eh = 0, mw = 0, mh = 0;
        for (size_t i = 0; i < a.size(); ++i) { double s = 0; for (int sp = 0; sp < spW; ++sp) s += gw[(size_t)sp * P * K + i]; ew = std::max(ew, std::fabs(s - a[i])); mw = std::max(mw, std::fabs(a[i])); }
        for (size_t i = 0; i < b.size(); ++i) { double s = 0; for (int sp = 0; sp < spH; ++sp) s += gh[(size_t)sp * K * N + i]; eh = std::max(eh, std::fabs(s - b[i])); mh = std::max(mh, std::fabs(b[i])); }
        printf("   check: W side max rel err %.2e, H side max rel err %.2e  %s\n", ew / mw, eh / mh, (ew / mw < 1e-4 && eh / mh < 1e-4) ? "OK" : "MISMATCH");
        CK(hipFree(rW)); CK(hipFree(rH));
    }
    CK(hipFree(W)); CK(hipFree(H)); CK(hipFree(X)); CK(hipFree(Xt)); CK(hipFree(outW)); CK(hipFree(outH));
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 8;
    run<64>(256, 512, 1, true);
    run<128>(512, 256, 1, true);
    run<256>(512, 768, 1, true);
    run<256>(16384, 16384, reps, false);
    run<256>(8192, 16384, reps, false);
    run<256>(16384, 8192, reps, false);
    run<64>(4096, 4096, reps, false);
    run<256>(16384, 2048, reps, false);
    return 0;
}
