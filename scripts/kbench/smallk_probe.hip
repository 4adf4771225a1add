// Development micro-benchmark: where do the 31 us of a small-k stripe kernel go?  smallk_h_kernel with parts of its main loop
// removed (template parameter PROBE, smallk.hpp).  Measured BEFORE the staging / fragment-read fixes of round 3 (MI355X, 4096 x 4096,
// event-timed: +4 us over the kernel's own duration; after them: full 34.7, no global loads 33.0, no MFMAs 27.7, no staging 26.3, registers only 21.9):
// full 35.8 us; no global loads in the loop 31.9; no MFMAs 35.2 (the MFMAs are entirely hidden); no staging 24.8; MFMAs on
// registers only 21.7 = the 13.7 us MFMA floor + 8 us of launch / prologue / epilogue.  So the loop is the staging path
// (global load -> VGPR -> ds_write -> barrier -> ds_read), not the matrix cores.  Two re-designs that attack it were built and
// measured slower (scripts/experiments/r03_smallk_paired_and_direct_fragments.patch): W fragments straight from global memory
// (16 lines per wave-load instead of 8: 46.6 us) and two workgroups per 32-wide stripe sharing the contraction (-40 % bytes per
// workgroup, +3-5 us per launch).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I nmf.jl_amd/csrc scripts/kbench/smallk_probe.hip -o scripts/kbench/smallk_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>
#include "smallk.hpp"
using namespace nmfx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int ST, int PROBE> float run(const float *X, int64_t P, int64_t N, const float *W, const float *G, const float *Ho, float *Hn, float *slabs, double *stat) {
    constexpr int lds = smallk_h_lds<ST>() * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&smallk_h_kernel<ST, PROBE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 30; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((smallk_h_kernel<ST, PROBE>), dim3((unsigned)(N / 16)), dim3(SMALLK_THREADS), lds, 0, X, P, P, W, G, Ho, Hn, 1e-4f, 1e-9f, slabs, stat, (const int *)nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 2) best = std::min(best, ms);
    }
    CK(hipGetLastError());
    return best * 1e3f;
}

int main(int argc, char **argv) {
    const int64_t P = argc > 1 ? atoll(argv[1]) : 4096, N = argc > 2 ? atoll(argv[2]) : 4096;
    std::vector<float> hx((size_t)P * N), hw((size_t)P * 64), hh((size_t)64 * N), hg(4096);
    srand(1);
    for (auto &v : hx) v = rand() / (float)RAND_MAX;
    for (auto &v : hw) v = rand() / (float)RAND_MAX;
    for (auto &v : hh) v = rand() / (float)RAND_MAX;
    for (auto &v : hg) v = rand() / (float)RAND_MAX;
    float *X, *W, *G, *Ho, *Hn, *slabs; double *stat;
    CK(hipMalloc(&X, hx.size() * 4)); CK(hipMalloc(&W, hw.size() * 4)); CK(hipMalloc(&G, 4096 * 4)); CK(hipMalloc(&Ho, hh.size() * 4));
    CK(hipMalloc(&Hn, hh.size() * 4)); CK(hipMalloc(&slabs, (size_t)(N / 16) * 4096 * 4)); CK(hipMalloc(&stat, (size_t)(N / 16) * 128 * 8));
    CK(hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(G, hg.data(), 4096 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(Ho, hh.data(), hh.size() * 4, hipMemcpyHostToDevice));
    printf("P=%lld N=%lld (MFMA floor %.1f us)\n", (long long)P, (long long)N, 2.0 * P * N * 64 / 157.3e6);
#define ROW(ST, PR, what) printf("ST=%3d %-58s %7.2f us\n", ST, what, run<ST, PR>(X, P, N, W, G, Ho, Hn, slabs, stat))
    ROW(128, 0, "full kernel");
    ROW(128, 1, "no global loads in the loop");
    ROW(128, 2, "no MFMAs (1 v_fma per fragment pair)");
    ROW(128, 3, "no staging (no LDS stores, no global loads)");
    ROW(128, 4, "MFMAs on registers: no staging, no fragment reads");
    ROW(64, 0, "full kernel");
    ROW(64, 1, "no global loads in the loop");
    ROW(64, 3, "no staging (no LDS stores, no global loads)");
    ROW(64, 4, "MFMAs on registers: no staging, no fragment reads");
    return 0;
}
