// Issue rate of v_mfma_f32_16x16x4_f32 as the small-k kernels use it: W waves per SIMD, C independent accumulator chains.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4 __attribute__((ext_vector_type(4)));
template <int CH> __global__ __launch_bounds__(1024) void probe(float *out, int iters, float a0, float b0) {
    v4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = v4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    v4 s = acc[0];
    for (int c = 1; c < CH; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <int CH> void run(int threads, float *d) {
    const int iters = 512 * 4 / CH / 8 * (512 / threads);   // same total MFMA count per SIMD for every configuration: 1024 per SIMD... scaled below
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int total_per_wave = 8 * CH * iters;
    hipLaunchKernelGGL(probe<CH>, dim3(256), dim3(threads), 0, 0, d, iters, 1.0f, 2.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(probe<CH>, dim3(256), dim3(threads), 0, 0, d, iters, 1.0f, 2.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)total_per_wave * (threads / 64) / 4.0;
    std::printf("threads %4d chains %d: %d MFMA/wave, %.0f MFMA/SIMD, %.2f us per launch -> %.1f cycles per MFMA per SIMD at 2.4 GHz\n", threads, CH, total_per_wave,
                per_simd, ms * 1000 / 20, ms * 1e-3 / 20 * 2.4e9 / per_simd);
}
int main() {
    float *d; hipMalloc(&d, 256 * 1024 * 4);
    run<4>(512, d); run<4>(256, d); run<4>(1024, d); run<2>(512, d); run<1>(512, d); run<8>(512, d); run<8>(256, d);
    return 0;
}
