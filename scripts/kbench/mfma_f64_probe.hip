// Issue rate of v_mfma_f64_16x16x4_f64 (the only dense Float64 matrix-core instruction the products of ALSPGrad use): W waves per SIMD,
// C independent accumulator chains, no memory traffic at all -- the ceiling any Float64 product can reach on this chip, to price the
// "0.76 of the 78.6 TFLOP/s peak" of the C5 trial-step products against.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/kbench/mfma_f64_probe.hip -o /tmp/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int CH> __global__ __launch_bounds__(1024) void probe(double *out, int iters, double a0, double b0) {
    d4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = d4{0., 0., 0., 0.};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    d4 s = acc[0];
    for (int c = 1; c < CH; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <int CH> void run(int threads, double *d, int blocks_per_cu) {
    const int iters = 4096 / CH;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const long long per_wave = 8LL * CH * iters;
    const int blocks = 256 * blocks_per_cu;
    hipLaunchKernelGGL(probe<CH>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0, 2.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(probe<CH>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0, 2.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double sec = ms * 1e-3 / 10;
    const double flops = (double)per_wave * (threads / 64) * blocks * 2048.0;
    const double per_simd = (double)per_wave * (threads / 64) * blocks_per_cu / 4.0;
    std::printf("threads %4d x %d block(s)/CU, %d chain(s): %.1f us per launch, %.1f TFLOP/s, %.1f cycles per MFMA per SIMD at 2.4 GHz\n", threads, blocks_per_cu, CH, sec * 1e6,
                flops / sec / 1e12, sec * 2.4e9 / per_simd);
}
// sustained load: the 3.5 ms launch back to back for ~1.5 s, throughput per window of 40 launches (does the clock give way under a long
// Float64 matrix-core load, as ALSPGrad's 200 ms outer iterations are?)
void sustained(double *d) {
    hipEvent_t e[12];
    for (auto &x : e) hipEventCreate(&x);
    const int iters = 1024, threads = 1024, blocks = 256;
    const double flops_per_launch = 8.0 * 4 * iters * (threads / 64) * blocks * 2048.0;
    hipEventRecord(e[0]);
    for (int w = 0; w < 11; ++w) {
        for (int r = 0; r < 40; ++r) hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0, 2.0);
        hipEventRecord(e[w + 1]);
    }
    hipEventSynchronize(e[11]);
    for (int w = 0; w < 11; ++w) {
        float ms; hipEventElapsedTime(&ms, e[w], e[w + 1]);
        std::printf("sustained window %2d (%.0f ms): %.1f TFLOP/s\n", w, ms, 40 * flops_per_launch / (ms * 1e-3) / 1e12);
    }
}
int main() {
    double *d; hipMalloc(&d, 512 * 1024 * 8);
    run<1>(256, d, 1); run<2>(256, d, 1); run<4>(256, d, 1); run<8>(256, d, 1);
    run<4>(512, d, 1); run<8>(512, d, 1); run<4>(1024, d, 1); run<4>(256, d, 2);
    sustained(d);
    return 0;
}
