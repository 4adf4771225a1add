// Development micro-benchmark: the persistent, cross-tile pipelined W*H kernel (gemm_stream.hpp) against the block-per-tile
// kernel (gemm_mfma.hpp) on the ratio / objective epilogues; checks that Q is bit-identical and the objective sums agree.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I nmf.jl_amd/csrc scripts/kbench/stream_bench.hip -o scripts/kbench/stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#include "gemm_mfma.hpp"
#include "gemm_stream.hpp"
using namespace nmfx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// experiment-only epilogues: where does the time of the stream kernel go?
template <typename T> struct SEpiNull {        // consumes the accumulators, touches no memory
    T *Q; int64_t ld; T acc0; LaneAddr<T> la;
    struct Tile { int dummy; };
    struct Pre {};
    struct St {};
    __device__ __forceinline__ void init(const TileCtx &t) { la.init(t, ld); acc0 = 0; }
    __device__ __forceinline__ Tile tile(int64_t, int64_t) const { return Tile{0}; }
    __device__ __forceinline__ Pre prefetch(const Tile &, int, int) const { return Pre{}; }
    template <int S> __device__ __forceinline__ void stage(const Tile &, int, int, T v, const Pre &, St &) { if constexpr (S == 0) { acc0 += v; asm volatile("" : "+v"(acc0)); } }
    __device__ __forceinline__ void finish(double *, int tid, int, int bid) { if (acc0 == 12345.f) Q[tid + bid] = acc0; }
};
template <typename T> struct SEpiStoreOnly {   // Q = acc
    T *Q; int64_t ld; LaneAddr<T> la;
    struct Tile { rsrc_t rq; };
    struct Pre {};
    struct St {};
    __device__ __forceinline__ void init(const TileCtx &t) { la.init(t, ld); }
    __device__ __forceinline__ Tile tile(int64_t rw0, int64_t cw0) const { Tile t; t.rq = __builtin_amdgcn_make_buffer_rsrc((void *)(Q + (cw0 + rw0 * ld)), 0, -1, 0x00020000); return t; }
    __device__ __forceinline__ Pre prefetch(const Tile &, int, int) const { return Pre{}; }
    template <int S> __device__ __forceinline__ void stage(const Tile &t, int ro, int co, T v, const Pre &, St &) { if constexpr (S == 2) buf_st(t.rq, la.lb, la.soff(ro, co), v); }
    __device__ __forceinline__ void finish(double *, int, int, int) {}
};
template <typename T> struct SEpiMulStore {   // Q = X * acc (load + store, no division)
    const T *X; T *Q; int64_t ld; LaneAddr<T> la;
    struct Tile { rsrc_t rx, rq; };
    struct Pre { T x; };
    struct St {};
    __device__ __forceinline__ void init(const TileCtx &t) { la.init(t, ld); }
    __device__ __forceinline__ Tile tile(int64_t rw0, int64_t cw0) const { Tile t; t.rx = __builtin_amdgcn_make_buffer_rsrc((void *)(X + (cw0 + rw0 * ld)), 0, -1, 0x00020000); t.rq = __builtin_amdgcn_make_buffer_rsrc((void *)(Q + (cw0 + rw0 * ld)), 0, -1, 0x00020000); return t; }
    __device__ __forceinline__ Pre prefetch(const Tile &t, int ro, int co) const { return Pre{buf_ld<T>(t.rx, la.lb, la.soff(ro, co))}; }
    template <int S> __device__ __forceinline__ void stage(const Tile &t, int ro, int co, T v, const Pre &p, St &) { if constexpr (S == 2) buf_st(t.rq, la.lb, la.soff(ro, co), p.x * v); }
    __device__ __forceinline__ void finish(double *, int, int, int) {}
};

template <typename F> float time_best(int reps, F &&f) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (i > 1) best = std::min(best, ms);
    }
    CK(hipGetLastError());
    return best;
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 12;
    const int64_t P = argc > 2 ? atoll(argv[2]) : 16384, N = argc > 3 ? atoll(argv[3]) : 16384, K = argc > 4 ? atoll(argv[4]) : 256;
    const int grid = argc > 5 ? atoi(argv[5]) : 512;
    float *H, *W, *X, *Q, *Q2; double *part, *part2;
    CK(hipMalloc(&H, K * N * 4)); CK(hipMalloc(&W, P * K * 4)); CK(hipMalloc(&X, P * N * 4)); CK(hipMalloc(&Q, P * N * 4)); CK(hipMalloc(&Q2, P * N * 4));
    CK(hipMalloc(&part, (size_t)(P / 128) * (N / 128) * 8)); CK(hipMalloc(&part2, 65536 * 8));
    std::vector<float> h((size_t)P * N);
    for (auto &v : h) v = (float)(rand() / (double)RAND_MAX);
    CK(hipMemcpy(X, h.data(), (size_t)P * N * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < (size_t)K * N; ++i) h[i] = (float)(rand() / (double)RAND_MAX) * 0.1f;
    CK(hipMemcpy(H, h.data(), (size_t)K * N * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < (size_t)P * K; ++i) h[i] = (float)(rand() / (double)RAND_MAX) * 0.1f;
    CK(hipMemcpy(W, h.data(), (size_t)P * K * 4, hipMemcpyHostToDevice));
    CK(hipMemset(Q, 0xff, (size_t)P * N * 4)); CK(hipMemset(Q2, 0xee, (size_t)P * N * 4));
    GemmArgs<float> g;
    g.A = H; g.lda = K; g.B = W; g.ldb = P; g.splits = 1; g.kchunk = (int)K; g.c_fastest = 1; g.done = nullptr;
    g.tiles_r = (int)(N / 128); g.tiles_c = (int)(P / 128);
    g.group = (g.tiles_r % 8 == 0 && g.tiles_c % 8 == 0) ? 8 : 1;
    const int blocks = g.tiles_r * g.tiles_c;
    StreamArgs s;
    s.A = H; s.lda = K; s.B = W; s.ldb = P; s.tiles_r = g.tiles_r; s.tiles_c = g.tiles_c; s.nkt = (int)(K / 32); s.group = g.group; s.done = nullptr;
    const double flop = 2.0 * P * N * K;
    auto report = [&](const char *name, float ms) { printf("%-36s %.1f us  %.1f TF/s\n", name, ms * 1e3, flop / (ms * 1e-3) / 1e12); fflush(stdout); };
    {
        SEpiNull<float> sn{Q2, P, 0.f};
        report("null   stream", time_best(reps, [&] { hipLaunchKernelGGL((gemm_wh_stream_kernel<SEpiNull<float>>), dim3(grid), dim3(256), 0, 0, s, sn); }));
        SEpiStoreOnly<float> ss{Q2, P};
        report("store  stream", time_best(reps, [&] { hipLaunchKernelGGL((gemm_wh_stream_kernel<SEpiStoreOnly<float>>), dim3(grid), dim3(256), 0, 0, s, ss); }));
        SEpiMulStore<float> sm{X, Q2, P};
        report("x*acc  stream", time_best(reps, [&] { hipLaunchKernelGGL((gemm_wh_stream_kernel<SEpiMulStore<float>>), dim3(grid), dim3(256), 0, 0, s, sm); }));
        EpiStore<float> es{Q, P, 0, nullptr};
        report("store  block-per-tile", time_best(reps, [&] { hipLaunchKernelGGL((gemm_mfma_kernel<float, KCONTIG, KSTRIDED, 128, 128, 2, 2, EpiStore<float>>), dim3(blocks), dim3(256), 0, 0, g, es); }));
    }
    {
        EpiRatio<float> er{X, Q, P, 3.4e-4f};
        report("ratio  block-per-tile", time_best(reps, [&] { hipLaunchKernelGGL((gemm_mfma_kernel<float, KCONTIG, KSTRIDED, 128, 128, 2, 2, EpiRatio<float>>), dim3(blocks), dim3(256), 0, 0, g, er); }));
        SEpiRatio<float> sr{X, Q2, P, 3.4e-4f};
        report("ratio  stream lead 1", time_best(reps, [&] { hipLaunchKernelGGL((gemm_wh_stream_kernel<SEpiRatio<float>, 1>), dim3(grid), dim3(256), 0, 0, s, sr); }));
        report("ratio  stream lead 3", time_best(reps, [&] { hipLaunchKernelGGL((gemm_wh_stream_kernel<SEpiRatio<float>, 3>), dim3(grid), dim3(256), 0, 0, s, sr); }));
        CK(hipMemset(Q2, 0xee, (size_t)P * N * 4));
        report("ratio  stream", time_best(reps, [&] { hipLaunchKernelGGL((gemm_wh_stream_kernel<SEpiRatio<float>>), dim3(grid), dim3(256), 0, 0, s, sr); }));
        std::vector<float> a((size_t)P * N), b((size_t)P * N);
        CK(hipMemcpy(a.data(), Q, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), Q2, b.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0, first = 0;
        for (size_t i = 0; i < a.size(); ++i) if (std::memcmp(&a[i], &b[i], 4) != 0) { if (!bad) first = i; ++bad; }
        printf("ratio: %zu of %zu elements differ", bad, a.size());
        if (bad) printf(" (first at row %zu col %zu: %.9g vs %.9g)", first % P, first / P, a[first], b[first]);
        printf("\n");
    }
    {
        EpiObjective<float, 0> eo{X, P, part, 0.0};
        report("sqdist block-per-tile", time_best(reps, [&] { hipLaunchKernelGGL((gemm_mfma_kernel<float, KCONTIG, KSTRIDED, 128, 128, 2, 2, EpiObjective<float, 0>>), dim3(blocks), dim3(256), 0, 0, g, eo); }));
        SEpiObjective<float, 0> so{X, P, part2, 0.0};
        report("sqdist stream", time_best(reps, [&] { hipLaunchKernelGGL((gemm_wh_stream_kernel<SEpiObjective<float, 0>>), dim3(grid), dim3(256), 0, 0, s, so); }));
        report("sqdist stream lead 1", time_best(reps, [&] { hipLaunchKernelGGL((gemm_wh_stream_kernel<SEpiObjective<float, 0>, 1>), dim3(grid), dim3(256), 0, 0, s, so); }));
        std::vector<double> pa(blocks), pb(grid);
        CK(hipMemcpy(pa.data(), part, pa.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(pb.data(), part2, pb.size() * 8, hipMemcpyDeviceToHost));
        double sa = 0, sb = 0; for (double v : pa) sa += v; for (double v : pb) sb += v;
        printf("sqdist: %.17g vs %.17g  rel diff %.3e\n", sa, sb, std::fabs(sa - sb) / std::fabs(sa));
    }
    {
        EpiObjective<float, 1> eo{X, P, part, 0.0};
        report("kldiv  block-per-tile", time_best(reps, [&] { hipLaunchKernelGGL((gemm_mfma_kernel<float, KCONTIG, KSTRIDED, 128, 128, 2, 2, EpiObjective<float, 1>>), dim3(blocks), dim3(256), 0, 0, g, eo); }));
        SEpiObjective<float, 1> so{X, P, part2, 0.0};
        report("kldiv  stream", time_best(reps, [&] { hipLaunchKernelGGL((gemm_wh_stream_kernel<SEpiObjective<float, 1>>), dim3(grid), dim3(256), 0, 0, s, so); }));
        std::vector<double> pa(blocks), pb(grid);
        CK(hipMemcpy(pa.data(), part, pa.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(pb.data(), part2, pb.size() * 8, hipMemcpyDeviceToHost));
        double sa = 0, sb = 0; for (double v : pa) sa += v; for (double v : pb) sb += v;
        printf("kldiv: %.17g vs %.17g  rel diff %.3e\n", sa, sb, std::fabs(sa - sb) / std::fabs(sa));
    }
    return 0;
}
