// Development micro-benchmark for the MFMA GEMM template at the bench shapes (not part of the product build).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I nmf.jl_amd/csrc scripts/kbench/gemm_bench.hip -o scripts/kbench/gemm_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "gemm_mfma.hpp"
using namespace nmfx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static int g_mode = 0, g_stagger = 0, g_prio = 0;
template <typename T, int LA, int LB, int BR, int BC, int WGR, int WGC, int BUF = 0>
double run(const char *name, int64_t R, int64_t C, int64_t Kd, int splits, bool c_fastest, int reps) {
    // A: R rows, B: C rows; KCONTIG -> ld = Kd, KSTRIDED -> ld = rows
    const int64_t lda = (LA == KCONTIG) ? Kd : R, ldb = (LB == KCONTIG) ? Kd : C;

    T *A, *B, *D;
    CK(hipMalloc(&A, (size_t)R * Kd * sizeof(T))); CK(hipMalloc(&B, (size_t)C * Kd * sizeof(T)));
    CK(hipMalloc(&D, (size_t)R * C * splits * sizeof(T))); CK(hipMemset(D, 0, (size_t)R * C * splits * sizeof(T)));
    std::vector<T> h((size_t)std::max(R, C) * Kd);
    // g_mode 0: signed U[-.5,.5) both; 1: U[0,1) both; 2: solver-like (big operand ~64*U, small operand ~U/16384)
    const bool a_big = R >= C;
    for (auto &v : h) { double u = rand() / (double)RAND_MAX; v = (T)(g_mode == 0 ? u - 0.5 : g_mode == 1 ? u : (a_big ? 64.0 * u : u / 16384.0)); }
    CK(hipMemcpy(A, h.data(), (size_t)R * Kd * sizeof(T), hipMemcpyHostToDevice));
    for (auto &v : h) { double u = rand() / (double)RAND_MAX; v = (T)(g_mode == 0 ? u - 0.5 : g_mode == 1 ? u : (a_big ? u / 16384.0 : 64.0 * u)); }
    CK(hipMemcpy(B, h.data(), (size_t)C * Kd * sizeof(T), hipMemcpyHostToDevice));
    GemmArgs<T> g;
    g.A = A; g.B = B; g.lda = lda; g.ldb = ldb; g.tiles_r = (int)(R / BR); g.tiles_c = (int)(C / BC);
    g.splits = splits; g.kchunk = (int)(Kd / splits); g.c_fastest = c_fastest; g.done = nullptr; g.prio = g_prio;
    EpiStore<T> e{D, C, R * C, nullptr};
    T *D2 = nullptr;
    if (g_stagger && C <= 1024) {   // Gram tail (only meaningful when the small operand is B): extra tiles A2 = B (the small operand), (C/BC)^2 tail tiles
        g.A2 = B; g.lda2 = ldb; g.r_split = R; g.tail_tiles = (int)(C / BR); g.tail_nkt = (int)(Kd / Mfma<T>::BK);
        const int blocks0 = g.tiles_r * g.tiles_c * splits;
        const int tt = g.tail_tiles * g.tiles_c;
        int per = std::max(1, (int)(((int64_t)tt * g.tail_nkt + blocks0 - 1) / blocks0));
        while ((int64_t)tt * ((g.tail_nkt + per - 1) / per) > blocks0) ++per;
        g.tail_per = per;
        const int pieces = (g.tail_nkt + per - 1) / per;
        CK(hipMalloc(&D2, (size_t)pieces * C * C * sizeof(T)));
        e.C2 = D2; e.ld2 = C; e.stride2 = C * C; e.r_off = R; e.c_off = 0;
    }
    const int blocks = g.tiles_r * g.tiles_c * splits;

    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((gemm_mfma_kernel<T, LA, LB, BR, BC, WGR, WGC, EpiStore<T>, 0, BUF>), dim3(blocks), dim3(WGR * WGC * 64), 0, 0, g, e);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (i > 0) best = std::min(best, ms);
    }
    CK(hipGetLastError());
    // spot check a few entries against fp64
    std::vector<T> hA((size_t)R * Kd), hB((size_t)C * Kd), hD((size_t)R * C * splits);
    CK(hipMemcpy(hA.data(), A, hA.size() * sizeof(T), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hB.data(), B, hB.size() * sizeof(T), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hD.data(), D, hD.size() * sizeof(T), hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int s = 0; s < 64; ++s) {
        const int64_t r = (rand() % R), c = (rand() % C);
        double ref = 0;
        for (int64_t kk = 0; kk < Kd; ++kk) {
            const double a = (LA == KCONTIG) ? hA[r * lda + kk] : hA[kk * lda + r];
            const double b = (LB == KCONTIG) ? hB[c * ldb + kk] : hB[kk * ldb + c];
            ref += a * b;
        }
        double got = 0;
        for (int sp = 0; sp < splits; ++sp) got += hD[(size_t)sp * R * C + c + r * C];
        maxerr = std::max(maxerr, std::fabs(got - ref) / (std::fabs(ref) + 1e-30));
    }
    const double tf = 2.0 * R * C * Kd / (best * 1e-3) / 1e12;
    printf("%-28s R=%lld C=%lld K=%lld splits=%d blocks=%d: %.1f us  %.1f TF/s  maxerr %.2e\n", name, (long long)R, (long long)C,
           (long long)Kd, splits, blocks, best * 1e3, tf, maxerr);
    fflush(stdout);
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(D));
    if (D2) CK(hipFree(D2));
    return tf;
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 6;
    const int what = argc > 2 ? atoi(argv[2]) : 0;
    g_mode = 1;
    if (what == 6) {   // the two big products without split-K: 512 half-size tiles instead of 256 x 2 splits (no slabs to combine)
        g_stagger = 1;
        for (int round = 0; round < 3; ++round) {
            printf("--- round %d\n", round);
            run<float, KCONTIG, KCONTIG, 128, 128, 2, 2, 2>("TN big 128x128 s2 unr2", 16384, 256, 16384, 2, true, reps);
            run<float, KCONTIG, KCONTIG, 128, 64, 4, 1, 1>("TN big 128x64 s1", 16384, 256, 16384, 1, true, reps);
            run<float, KCONTIG, KCONTIG, 64, 128, 1, 4, 1>("TN big 64x128 s1", 16384, 256, 16384, 1, true, reps);
            run<float, KSTRIDED, KSTRIDED, 128, 128, 2, 2, 1>("NT big 128x128 s2", 256, 16384, 16384, 2, false, reps);
            run<float, KSTRIDED, KSTRIDED, 128, 64, 4, 1, 1>("NT big 128x64 s1", 256, 16384, 16384, 1, false, reps);
            run<float, KSTRIDED, KSTRIDED, 64, 128, 1, 4, 1>("NT big 64x128 s1", 256, 16384, 16384, 1, false, reps);
        }
        return 0;
    }
    if (what == 8) {   // Float64: where does the distance to the instruction's 77.5 TFLOP/s sit -- per k-tile (slope over the contraction) or per output tile (offset)?
        for (int round = 0; round < 2; ++round) {
            printf("--- round %d\n", round);
            for (int64_t kd : {512, 1024, 2048, 4096}) {
                run<double, KSTRIDED, KSTRIDED, 128, 64, 4, 1, 1>("f64 W-side 128x64", 512, 32768, kd, 1, false, reps);
                run<double, KSTRIDED, KSTRIDED, 64, 128, 1, 4, 1>("f64 W-side 64x128", 512, 32768, kd, 1, false, reps);
                run<double, KSTRIDED, KSTRIDED, 128, 128, 2, 2, 1>("f64 W-side 128x128", 512, 32768, kd, 1, false, reps);
            }
            run<double, KCONTIG, KCONTIG, 64, 64, 2, 2, 1>("f64 H-side 64x64", 4096, 512, 512, 1, true, reps);
            run<double, KCONTIG, KCONTIG, 128, 64, 4, 1, 1>("f64 H-side 128x64", 4096, 512, 512, 1, true, reps);
            run<double, KCONTIG, KCONTIG, 128, 128, 2, 2, 1>("f64 WtX shard 128x128 s2", 4096, 512, 32768, 2, true, reps);
            run<double, KCONTIG, KCONTIG, 128, 64, 4, 1, 1>("f64 WtX shard 128x64 s2", 4096, 512, 32768, 2, true, reps);
            run<double, KCONTIG, KCONTIG, 128, 64, 4, 1, 1>("f64 WtX shard 128x64 s1", 4096, 512, 32768, 1, true, reps);
        }
        return 0;
    }
    if (what == 5) {   // alspgrad's Float64 trial-step shapes (C5 shard) with a PLAIN store: what the tilings reach without the fused loader / epilogue
        for (int round = 0; round < 2; ++round) {
            printf("--- round %d\n", round);
            // W side: Gram (512 x 512) x Z' (32768 rows): R = 512, C = 32768, contraction 512, both k-strided
            run<double, KSTRIDED, KSTRIDED, 128, 64, 4, 1, 1>("f64 W-side 128x64", 512, 32768, 512, 1, false, reps);
            run<double, KSTRIDED, KSTRIDED, 64, 128, 1, 4, 1>("f64 W-side 64x128", 512, 32768, 512, 1, false, reps);
            run<double, KSTRIDED, KSTRIDED, 128, 128, 2, 2, 1>("f64 W-side 128x128", 512, 32768, 512, 1, false, reps);
            run<double, KSTRIDED, KSTRIDED, 64, 64, 2, 2, 1>("f64 W-side 64x64", 512, 32768, 512, 1, false, reps);
            // H side: Z (512 x 4096) with Gram: R = 4096, C = 512, contraction 512, both k-contiguous
            run<double, KCONTIG, KCONTIG, 64, 64, 2, 2, 1>("f64 H-side 64x64", 4096, 512, 512, 1, true, reps);
            run<double, KCONTIG, KCONTIG, 64, 128, 1, 4, 1>("f64 H-side 64x128", 4096, 512, 512, 1, true, reps);
            run<double, KCONTIG, KCONTIG, 128, 64, 4, 1, 1>("f64 H-side 128x64", 4096, 512, 512, 1, true, reps);
            // the p*n*k product of the shard for comparison (contraction 4096)
            run<double, KCONTIG, KCONTIG, 128, 128, 2, 2, 1>("f64 WtX shard", 4096, 512, 32768, 2, true, reps);
        }
        return 0;
    }
    if (what == 4) {   // the 8-rank shard shapes: tile shapes / splits with buffer-load staging
        g_stagger = 1;
        for (int round = 0; round < 3; ++round) {
            printf("--- round %d\n", round);
            run<float, KSTRIDED, KSTRIDED, 128, 128, 2, 2, 1>("NT sh8 128x128 s1", 256, 16384, 2048, 1, false, reps);
            run<float, KSTRIDED, KSTRIDED, 128, 128, 2, 2, 1>("NT sh8 128x128 s2", 256, 16384, 2048, 2, false, reps);
            run<float, KSTRIDED, KSTRIDED, 128, 64, 4, 1, 1>("NT sh8 128x64 s1", 256, 16384, 2048, 1, false, reps);
            run<float, KSTRIDED, KSTRIDED, 64, 128, 1, 4, 1>("NT sh8 64x128 s1", 256, 16384, 2048, 1, false, reps);
            run<float, KCONTIG, KCONTIG, 128, 128, 2, 2, 2>("TN sh8 128x128 s16", 2048, 256, 16384, 16, true, reps);
            run<float, KCONTIG, KCONTIG, 128, 128, 2, 2, 2>("TN sh8 128x128 s8", 2048, 256, 16384, 8, true, reps);
            run<float, KCONTIG, KCONTIG, 128, 64, 4, 1, 1>("TN sh8 128x64 s8", 2048, 256, 16384, 8, true, reps);
            run<float, KCONTIG, KCONTIG, 64, 128, 1, 4, 1>("TN sh8 64x128 s8", 2048, 256, 16384, 8, true, reps);
        }
        return 0;
    }
    if (what == 7) {   // 8-rank shard shapes: 4 waves (one per SIMD) against 8 waves (two per SIMD) on the same 128 x 128 tile
        for (int round = 0; round < 3; ++round) {
            printf("--- round %d\n", round);
            run<float, KCONTIG, KCONTIG, 128, 128, 2, 2, 2>("TN sh8 WtX 4 waves s8", 2048, 256, 16384, 8, true, reps);
            run<float, KCONTIG, KCONTIG, 128, 128, 2, 4, 2>("TN sh8 WtX 8 waves 2x4 s8", 2048, 256, 16384, 8, true, reps);
            run<float, KCONTIG, KCONTIG, 128, 128, 4, 2, 2>("TN sh8 WtX 8 waves 4x2 s8", 2048, 256, 16384, 8, true, reps);
            run<float, KCONTIG, KCONTIG, 128, 128, 2, 2, 2>("TN sh8 XHt' 4 waves s1", 256, 16384, 2048, 1, false, reps);
            run<float, KCONTIG, KCONTIG, 128, 128, 2, 4, 2>("TN sh8 XHt' 8 waves 2x4 s1", 256, 16384, 2048, 1, false, reps);
            run<float, KCONTIG, KCONTIG, 128, 128, 4, 2, 2>("TN sh8 XHt' 8 waves 4x2 s1", 256, 16384, 2048, 1, false, reps);
            run<float, KCONTIG, KCONTIG, 128, 128, 2, 2, 2>("TN sh8 XHt' 4 waves s2", 256, 16384, 2048, 2, false, reps);
        }
        return 0;
    }
    if (what == 3) {   // buffer loads vs buffer loads + k-loop unrolled by two (LDS stage offsets as immediates)
        g_stagger = 1;
        for (int round = 0; round < 4; ++round) {
            printf("--- round %d\n", round);
            run<float, KCONTIG, KCONTIG, 128, 128, 2, 2, 1>("TN big (WtX) buf", 16384, 256, 16384, 2, true, reps);
            run<float, KCONTIG, KCONTIG, 128, 128, 2, 2, 2>("TN big (WtX) buf+unr2", 16384, 256, 16384, 2, true, reps);
            run<float, KSTRIDED, KSTRIDED, 128, 128, 2, 2, 1>("NT big (XHt) buf", 256, 16384, 16384, 2, false, reps);
            run<float, KSTRIDED, KSTRIDED, 128, 128, 2, 2, 2>("NT big (XHt) buf+unr2", 256, 16384, 16384, 2, false, reps);
            run<float, KCONTIG, KCONTIG, 128, 128, 2, 2, 1>("TN shard/8 buf", 2048, 256, 16384, 16, true, reps);
            run<float, KCONTIG, KCONTIG, 128, 128, 2, 2, 2>("TN shard/8 buf+unr2", 2048, 256, 16384, 16, true, reps);
            run<double, KCONTIG, KCONTIG, 128, 128, 2, 2, 1>("TN f64 buf", 8192, 256, 8192, 4, true, reps);
            run<double, KCONTIG, KCONTIG, 128, 128, 2, 2, 2>("TN f64 buf+unr2", 8192, 256, 8192, 4, true, reps);
        }
        return 0;
    }
    if (what == 2) {   // pointer loads vs buffer loads with loop-invariant lane offsets (A/B inside one process, interleaved)
        g_stagger = 1;
        for (int round = 0; round < 4; ++round) {
            printf("--- round %d\n", round);
            run<float, KCONTIG, KCONTIG, 128, 128, 2, 2, 0>("TN big (WtX) ptr", 16384, 256, 16384, 2, true, reps);
            run<float, KCONTIG, KCONTIG, 128, 128, 2, 2, 1>("TN big (WtX) buf", 16384, 256, 16384, 2, true, reps);
            run<float, KSTRIDED, KSTRIDED, 128, 128, 2, 2, 0>("NT big (XHt) ptr", 256, 16384, 16384, 2, false, reps);
            run<float, KSTRIDED, KSTRIDED, 128, 128, 2, 2, 1>("NT big (XHt) buf", 256, 16384, 16384, 2, false, reps);
            run<float, KCONTIG, KCONTIG, 128, 128, 2, 2, 0>("TN shard/8 ptr", 2048, 256, 16384, 16, true, reps);
            run<float, KCONTIG, KCONTIG, 128, 128, 2, 2, 1>("TN shard/8 buf", 2048, 256, 16384, 16, true, reps);
            run<float, KSTRIDED, KSTRIDED, 128, 128, 2, 2, 0>("NT shard/8 ptr", 256, 16384, 2048, 1, false, reps);
            run<float, KSTRIDED, KSTRIDED, 128, 128, 2, 2, 1>("NT shard/8 buf", 256, 16384, 2048, 1, false, reps);
            run<double, KCONTIG, KCONTIG, 128, 128, 2, 2, 0>("TN f64 ptr", 8192, 256, 8192, 4, true, reps);
            run<double, KCONTIG, KCONTIG, 128, 128, 2, 2, 1>("TN f64 buf", 8192, 256, 8192, 4, true, reps);
            run<double, KSTRIDED, KSTRIDED, 128, 128, 2, 2, 0>("NT f64 ptr", 256, 8192, 8192, 4, false, reps);
            run<double, KSTRIDED, KSTRIDED, 128, 128, 2, 2, 1>("NT f64 buf", 256, 8192, 8192, 4, false, reps);
        }
        return 0;
    }
    if (what == 1) {   // wave-priority patterns on the two big products and their 8-rank shard shapes (A/B inside one process, interleaved)
        g_stagger = 1;
        for (int round = 0; round < 3; ++round)
            for (int pr : {0, 1, 2, 4, 3}) {
                g_prio = pr;
                printf("--- prio pattern %d (round %d)\n", pr, round);
                run<float, KCONTIG, KCONTIG, 128, 128, 2, 2>("TN big (WtX) 128x128", 16384, 256, 16384, 2, true, reps);
                run<float, KSTRIDED, KSTRIDED, 128, 128, 2, 2>("NT big (XHt) 128x128", 256, 16384, 16384, 2, false, reps);
                run<float, KCONTIG, KCONTIG, 128, 128, 2, 2>("TN shard/8 128x128", 2048, 256, 16384, 16, true, reps);
                run<float, KSTRIDED, KSTRIDED, 128, 128, 2, 2>("NT shard/8 128x128 no split", 256, 16384, 2048, 1, false, reps);
            }
        return 0;
    }
    for (int st : {0, 1}) {
    printf("--- gram tail %d\n", st);
    g_stagger = st;
    
    run<float, KCONTIG, KCONTIG, 128, 128, 2, 2>("TN big (WtX) 128x128", 16384, 256, 16384, 2, true, reps);
    run<float, KCONTIG, KCONTIG, 128, 128, 2, 2>("TN big+gram rows (260 tiles)", 16384 + 256, 256, 16384, 2, true, reps);
    run<float, KSTRIDED, KSTRIDED, 128, 128, 2, 2>("NT big (XHt) 128x128", 256, 16384, 16384, 2, false, reps);
    run<float, KCONTIG, KCONTIG, 128, 128, 2, 2>("TN shard/8 128x128", 2048, 256, 16384, 16, true, reps);
    run<float, KSTRIDED, KSTRIDED, 128, 128, 2, 2>("NT shard/8 128x128", 256, 16384, 2048, 2, false, reps);
    run<float, KSTRIDED, KSTRIDED, 128, 128, 2, 2>("NT shard/8 128x128 no split", 256, 16384, 2048, 1, false, reps);
    run<double, KCONTIG, KCONTIG, 128, 128, 2, 2>("TN f64 128x128", 8192, 256, 8192, 4, true, reps);
    run<double, KSTRIDED, KSTRIDED, 128, 128, 2, 2>("NT f64 128x128", 256, 8192, 8192, 4, false, reps);
    }
    return 0;
}
