// greedy_bench.hip -- development aid: GreedyCD's W-side sweep (cd.hpp) on a state written by greedy_state.py.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../nmf.jl_amd/csrc -I../../include greedy_bench.hip -o greedy_bench
//   ./greedy_bench state.bin
// Prints the launch time of the product kernel, executed steps, a checksum of the swept W, and the histogram of steps per row
// (how much of the launch is the tail of its longest rows).
#include "cd.hpp"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>
using namespace nmfx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <typename T, int KMAX>
__global__ __launch_bounds__(256) void sweep_rowsteps_kernel(SampleView<const T> Wold, SampleView<T> Wout, SampleView<const T> G, const T *__restrict__ P, int64_t ldp,
                                                             int64_t nsamples, int k, T lambda, T epsT, const T *pinit, int *row_steps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    if (i >= nsamples) return;
    auto fetch = [&](int q, int m) { return P[(int64_t)q * ldp + lane + 64 * m]; };
    const long long s = greedy_sweep_row<T, KMAX, true>(Wold, Wout, G, P, ldp, i, k, lambda, epsT, pinit, lane, fetch);
    if (lane == 0) row_steps[i] = (int)s;
}

int main(int argc, char **argv) {
    if (argc < 2) { printf("usage: greedy_bench state.bin\n"); return 1; }
    FILE *f = fopen(argv[1], "rb"); if (!f) { printf("cannot open %s\n", argv[1]); return 1; }
    int64_t hdr[2]; if (fread(hdr, 8, 2, f) != 2) return 1;
    const int64_t p = hdr[0]; const int k = (int)hdr[1];
    std::vector<float> W((size_t)p * k), G((size_t)p * k), Pm((size_t)k * k);
    if (fread(W.data(), 4, W.size(), f) != W.size() || fread(G.data(), 4, G.size(), f) != G.size() || fread(Pm.data(), 4, Pm.size(), f) != Pm.size()) return 1;
    fclose(f);
    constexpr int KMAX = 4;
    if (k != 64 * KMAX) { printf("built for k = %d\n", 64 * KMAX); return 1; }
    float *dW, *dWn, *dG, *dP, *part, *pinit; int *rs; long long *tot; int *done;
    CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dWn, W.size() * 4)); CK(hipMalloc(&dG, G.size() * 4)); CK(hipMalloc(&dP, Pm.size() * 4));
    CK(hipMalloc(&part, (p / 4 + 16) * 4)); CK(hipMalloc(&pinit, 64)); CK(hipMalloc(&rs, p * 4)); CK(hipMalloc(&tot, 8)); CK(hipMalloc(&done, 4));
    CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dG, G.data(), G.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dP, Pm.data(), Pm.size() * 4, hipMemcpyHostToDevice)); CK(hipMemset(done, 0, 4)); CK(hipMemset(tot, 0, 8));
    const SampleView<const float> Wo{dW, 1, p}, Gv{dG, 1, p}; const SampleView<float> Wn{dWn, 1, p};
    const unsigned blocks = (unsigned)((p + 3) / 4);
    const float epsT = std::numeric_limits<float>::epsilon();
    hipLaunchKernelGGL((greedy_pinit_kernel<float, KMAX>), dim3(blocks), dim3(256), 0, 0, Wo, Gv, dP, (int64_t)k, p, k, 0.f, epsT, part, done);
    int *queue; CK(hipMalloc(&queue, GREEDY_NQ * GREEDY_QSTRIDE * 4)); CK(hipMemset(queue, 0, GREEDY_NQ * GREEDY_QSTRIDE * 4));
    hipLaunchKernelGGL(greedy_pinit_reduce_kernel<float>, dim3(1), dim3(256), 0, 0, part, (int)blocks, pinit, queue, done);
    int per_cu = 1; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(&greedy_sweep_kernel<float, KMAX, true>), 256, 0));
    const unsigned pblocks = std::min<unsigned>(blocks, (unsigned)per_cu * 256u); printf("resident blocks per CU %d -> %u blocks\n", per_cu, pblocks);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) {
        CK(hipMemset(tot, 0, 8)); CK(hipMemset(queue, 0, GREEDY_NQ * GREEDY_QSTRIDE * 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((greedy_sweep_kernel<float, KMAX, true>), dim3(pblocks), dim3(256), 0, 0, Wo, Wn, Gv, dP, (int64_t)k, p, k, 0.f, epsT, pinit, queue, tot, done);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r > 0) best = std::min(best, ms);
    }
    long long steps; CK(hipMemcpy(&steps, tot, 8, hipMemcpyDeviceToHost));
    std::vector<float> Wh(W.size()); CK(hipMemcpy(Wh.data(), dWn, W.size() * 4, hipMemcpyDeviceToHost));
    unsigned long long h = 1469598103934665603ull; for (size_t i = 0; i < Wh.size(); ++i) { unsigned u; memcpy(&u, &Wh[i], 4); h = (h ^ u) * 1099511628211ull; }
    float pi; CK(hipMemcpy(&pi, pinit, 4, hipMemcpyDeviceToHost));
    printf("p %lld k %d  p_init %.6g  sweep %.3f ms  steps %lld  %.2f G steps/s  W checksum %016llx\n", (long long)p, k, pi, best, steps, steps / best / 1e6, h);
    // chain latency: one wave per SIMD (1024 rows), and one wave per CU (256 rows in 256 blocks would need a 64-thread launch: use rows 0, 4, 8...)
    for (int64_t rows : {(int64_t)1024, (int64_t)2048, (int64_t)4096, (int64_t)8192}) {
        if (rows > p) break;
        hipLaunchKernelGGL((sweep_rowsteps_kernel<float, KMAX>), dim3((unsigned)(rows / 4)), dim3(256), 0, 0, Wo, Wn, Gv, dP, (int64_t)k, rows, k, 0.f, epsT, pinit, rs);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((sweep_rowsteps_kernel<float, KMAX>), dim3((unsigned)(rows / 4)), dim3(256), 0, 0, Wo, Wn, Gv, dP, (int64_t)k, rows, k, 0.f, epsT, pinit, rs);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<int> rr((size_t)rows); CK(hipMemcpy(rr.data(), rs, rows * 4, hipMemcpyDeviceToHost));
        long long tot2 = 0; int mx = 0; for (int v : rr) { tot2 += v; mx = std::max(mx, v); }
        printf("first %5lld rows (%.0f waves per SIMD): %.3f ms, longest row %d steps -> %.0f ns per step of that row; %.2f G steps/s\n", (long long)rows, rows / 1024.0, ms, mx, ms * 1e6 / mx, tot2 / ms / 1e6);
    }
    hipLaunchKernelGGL((sweep_rowsteps_kernel<float, KMAX>), dim3(blocks), dim3(256), 0, 0, Wo, Wn, Gv, dP, (int64_t)k, p, k, 0.f, epsT, pinit, rs);
    std::vector<int> r((size_t)p); CK(hipMemcpy(r.data(), rs, p * 4, hipMemcpyDeviceToHost));
    std::vector<int> s = r; std::sort(s.begin(), s.end());
    long long sum = 0; for (int v : s) sum += v;
    printf("steps per row: mean %.1f  min %d  p50 %d  p90 %d  p99 %d  p99.9 %d  max %d\n", (double)sum / p, s[0], s[p / 2], s[p * 9 / 10], s[p * 99 / 100], s[p * 999 / 1000], s[p - 1]);
    // the longest chain a CU slot sees with the in-order block dispatch: blocks of 4 consecutive rows
    long long worst_block = 0; for (int64_t b = 0; b < p / 4; ++b) { long long m = 0; for (int j = 0; j < 4; ++j) m = std::max<long long>(m, r[b * 4 + j]); worst_block = std::max(worst_block, m); }
    printf("ns per step of the longest row if it alone set the launch time: %.1f\n", best * 1e6 / s[p - 1]);
    return 0;
}
