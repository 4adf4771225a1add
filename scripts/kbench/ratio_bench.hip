// Development micro-benchmark: the short-K, large-output GEMMs (W*H with ratio / objective / plain-store epilogues).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_mfma.hpp"
using namespace nmfx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int LB, int BR, int BC, int WGR, int WGC, typename Epi>
void run_upd(const char *name, GemmArgs<float> g, Epi e, int64_t R, int64_t C, int64_t Kd, int reps) {
    g.tiles_r = (int)(R / BR); g.tiles_c = (int)(C / BC); g.group = 1;
    const int blocks = g.tiles_r * g.tiles_c;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((gemm_mfma_kernel<float, KCONTIG, LB, BR, BC, WGR, WGC, Epi>), dim3(blocks), dim3(WGR * WGC * 64), 0, 0, g, e);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (i > 1) best = std::min(best, ms);
    }
    CK(hipGetLastError());
    printf("%-40s blocks=%6d: %.1f us  %.1f TF/s\n", name, blocks, best * 1e3, 2.0 * R * C * Kd / (best * 1e-3) / 1e12);
}

template <int BR, int BC, int WGR, int WGC, typename Epi>
void run(const char *name, GemmArgs<float> g, Epi e, int64_t R, int64_t C, int64_t Kd, int group, int reps) {
    g.tiles_r = (int)(R / BR); g.tiles_c = (int)(C / BC); g.group = group;
    const int blocks = g.tiles_r * g.tiles_c;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((gemm_mfma_kernel<float, KCONTIG, KSTRIDED, BR, BC, WGR, WGC, Epi>), dim3(blocks), dim3(WGR * WGC * 64), 0, 0, g, e);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (i > 1) best = std::min(best, ms);
    }
    CK(hipGetLastError());
    printf("%-40s blocks=%6d group=%d: %.1f us  %.1f TF/s\n", name, blocks, group, best * 1e3, 2.0 * R * C * Kd / (best * 1e-3) / 1e12);
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 12;
    const int64_t P = 16384, N = 16384, K = 256;
    float *H, *W, *X, *Q; double *part;
    CK(hipMalloc(&H, K * N * 4)); CK(hipMalloc(&W, P * K * 4)); CK(hipMalloc(&X, P * N * 4)); CK(hipMalloc(&Q, P * N * 4));
    CK(hipMalloc(&part, 8 * 65536 * 8));
    std::vector<float> h((size_t)P * N);
    for (auto &v : h) v = (float)(rand() / (double)RAND_MAX);
    CK(hipMemcpy(X, h.data(), (size_t)P * N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(H, h.data(), (size_t)K * N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data(), (size_t)P * K * 4, hipMemcpyHostToDevice));
    GemmArgs<float> g;
    g.A = H; g.lda = K; g.B = W; g.ldb = P; g.splits = 1; g.kchunk = (int)K; g.c_fastest = 1; g.done = nullptr;
    EpiStore<float> es{Q, P, 0, nullptr};
    EpiRatio<float> er{X, Q, P, 3.4e-4f};
    EpiObjective<float, 0> eo{X, P, part, 0.0};
    for (int group : {1, 8}) {
        run<128, 128, 2, 2>("store   128x128", g, es, N, P, K, group, reps);
        run<128, 128, 2, 2>("ratio   128x128", g, er, N, P, K, group, reps);
        run<128, 128, 2, 2>("sqdist  128x128", g, eo, N, P, K, group, reps);
    }
    {   // H update: out (n x k, ld = k) = H' (n x k) * Gram (k x k), numerator in 2 slabs
        float *Gm, *num, *Hn; double *st;
        CK(hipMalloc(&Gm, K * K * 4)); CK(hipMalloc(&num, 2 * K * N * 4)); CK(hipMalloc(&Hn, K * N * 4)); CK(hipMalloc(&st, 1024 * 256 * 2 * 8));
        CK(hipMemcpy(Gm, h.data(), (size_t)K * K * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(num, h.data(), (size_t)2 * K * N * 4, hipMemcpyHostToDevice));
        GemmArgs<float> u;
        u.A = H; u.lda = K; u.B = Gm; u.ldb = K; u.splits = 1; u.kchunk = (int)K; u.c_fastest = 1; u.done = nullptr;
        EpiStore<float> us{Hn, K, 0, nullptr};
        EpiMultUpdate<float, 1> um{num, 2, K * N, H, Hn, K, 0.0f, 3.4e-4f, st, (int)K};
        EpiMultUpdate<float, 0> um0{num, 2, K * N, H, Hn, K, 0.0f, 3.4e-4f, st, (int)K};
        EpiMultUpdate<float, 0> um1{num, 1, K * N, H, Hn, K, 0.0f, 3.4e-4f, st, (int)K};
        run_upd<KCONTIG, 64, 128, 1, 4>("updH store       64x128", u, us, N, K, K, reps);
        run_upd<KCONTIG, 64, 128, 1, 4>("updH mult+stats  64x128", u, um, N, K, K, reps);
        run_upd<KCONTIG, 64, 128, 1, 4>("updH mult        64x128", u, um0, N, K, K, reps);
        run_upd<KCONTIG, 64, 128, 1, 4>("updH mult 1 slab 64x128", u, um1, N, K, K, reps);
        run_upd<KCONTIG, 64, 64, 2, 2>("updH store       64x64", u, us, N, K, K, reps);
        run_upd<KCONTIG, 64, 64, 2, 2>("updH mult+stats  64x64", u, um, N, K, K, reps);
        run_upd<KCONTIG, 128, 128, 2, 2>("updH store       128x128", u, us, N, K, K, reps);
        run_upd<KCONTIG, 128, 128, 2, 2>("updH mult+stats  128x128", u, um, N, K, K, reps);
        run_upd<KCONTIG, 128, 64, 4, 1>("updH mult+stats  128x64", u, um, N, K, K, reps);
    }
    run<64, 128, 1, 4>("ratio   64x128 (1x4 waves)", g, er, N, P, K, 1, reps);
    run<128, 64, 4, 1>("ratio   128x64 (4x1 waves)", g, er, N, P, K, 1, reps);
    run<256, 64, 4, 1>("ratio   256x64 (4x1 waves)", g, er, N, P, K, 1, reps);
    run<64, 256, 1, 4>("ratio   64x256 (1x4 waves)", g, er, N, P, K, 1, reps);
    return 0;
}
