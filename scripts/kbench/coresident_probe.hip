// Round 6 probe: when does the workgroup scheduler place a side-stream workgroup on a CU that already holds a block of a resident
// kernel?  A hog kernel (one or two blocks per CU, 256 threads, 64 KiB of LDS, ~600 us of dependent arithmetic -- no memory traffic)
// runs on one stream; a guest workgroup (threads, registers, LDS given) is launched on a second (high-priority) stream a few us later
// and reports when its first instruction ran and how long a fixed piece of work took.
// build: hipcc --offload-arch=gfx950 -O3 -o coresident_probe coresident_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void hog(long long *t, int iters, float *sink, int lds_words) {
    extern __shared__ float sm[];
    if (threadIdx.x == 0) t[blockIdx.x] = (long long)wall_clock64();
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) { a = a * b + 0.5f; b = b * 0.99999f + 1e-6f; }
    if (lds_words > 0) sm[threadIdx.x % lds_words] = a;
    if (a == 12345.678f) sink[0] = a + sm[0];
    if (threadIdx.x == 0) t[gridDim.x + blockIdx.x] = (long long)wall_clock64();
}
template <int REGS> __global__ __launch_bounds__(256) void hog_regs(long long *t, int iters, float *sink) {
    __shared__ float sm[16384];   // 64 KiB static, as the product kernel's staging buffers
    if (threadIdx.x == 0) t[blockIdx.x] = (long long)wall_clock64();
    float v[REGS];
#pragma unroll
    for (int r = 0; r < REGS; ++r) v[r] = threadIdx.x + r;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < REGS; ++r) v[r] = v[r] * 1.0001f + 0.25f;
    }
    float s = 0;
#pragma unroll
    for (int r = 0; r < REGS; ++r) s += v[r];
    sm[threadIdx.x] = s;
    __syncthreads();
    if (s == 1.5f) sink[0] = s + sm[(threadIdx.x * 7) % 16384];
    if (threadIdx.x == 0) t[gridDim.x + blockIdx.x] = (long long)wall_clock64();
}
// MFMA-saturating hog: four independent accumulator tiles, back-to-back v_mfma_f32_32x32x2_f32 (64 cycles each on the SIMD's matrix pipe).
// YIELD: 0 nothing; 1 one s_nop per 16 MFMAs; 2 one s_sleep 1 per 16 MFMAs; 3 s_setprio 0 / raise around each group (no-op at prio 0)
typedef float f16v __attribute__((ext_vector_type(16)));
template <int YIELD> __global__ __launch_bounds__(256) void hog_mfma(long long *t, int iters, float *sink) {
    __shared__ float sm[16384];
    if (threadIdx.x == 0) t[blockIdx.x] = (long long)wall_clock64();
    f16v a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = 1.0f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
        }
        if (YIELD == 1) asm volatile("s_nop 0");
        if (YIELD == 2) __builtin_amdgcn_s_sleep(1);
    }
    float s = a0[0] + a1[1] + a2[2] + a3[3];
    sm[threadIdx.x] = s;
    __syncthreads();
    if (s == 1.5f) sink[0] = s + sm[(threadIdx.x * 7) % 16384];
    if (threadIdx.x == 0) t[gridDim.x + blockIdx.x] = (long long)wall_clock64();
}
template <int REGS> __global__ __launch_bounds__(512) void guest(long long *t, float *sink) {
    extern __shared__ float sm[];
    if (t[4] != 0) __builtin_amdgcn_s_setprio(3);
    if (threadIdx.x == 0) t[0] = (long long)wall_clock64();
    float v[REGS];
#pragma unroll
    for (int r = 0; r < REGS; ++r) v[r] = threadIdx.x + r;
    for (int i = 0; i < 2000; ++i) {
#pragma unroll
        for (int r = 0; r < REGS; ++r) v[r] = v[r] * 1.0001f + 0.25f;
    }
    float s = 0;
#pragma unroll
    for (int r = 0; r < REGS; ++r) s += v[r];
    sm[threadIdx.x] = s;
    if (s == 1.5f) sink[0] = s;
    if (threadIdx.x == 0) t[1] = (long long)wall_clock64();
}
int main() {
    int dev = 0; CK(hipSetDevice(dev));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, dev));
    const int ncu = pr.multiProcessorCount;
    long long *th, *tg; float *sink;
    CK(hipMalloc(&th, sizeof(long long) * 4096)); CK(hipMalloc(&tg, sizeof(long long) * 8)); CK(hipMalloc(&sink, 64));
    hipStream_t s0, s1; int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&hog), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&guest<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&guest<96>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t ev; CK(hipEventCreate(&ev));
    const int iters = 260000;   // ~600 us
    long long guest_prio = 0;
    int hog_kind = 0;   // 0: few registers, dynamic LDS; 1: 80 registers + 64 KiB static LDS; 2: 120 registers + 64 KiB static
    auto run = [&](int hog_blocks, int hog_lds_kib, int g_threads, int g_regs, int g_lds_kib, bool order_guest_first) -> int {
        CK(hipMemset(tg, 0, 64)); CK(hipMemcpy(tg + 4, &guest_prio, 8, hipMemcpyHostToDevice)); CK(hipDeviceSynchronize());
        // a small kernel first, so that both launches below hang off the same event (as the fork in enqueue_projals does)
        hipLaunchKernelGGL(hog, dim3(1), dim3(256), 1024, s0, th + 2048, 10, sink, 1);
        CK(hipEventRecord(ev, s0)); CK(hipStreamWaitEvent(s1, ev, 0));
        if (order_guest_first) {
            if (g_regs <= 16) hipLaunchKernelGGL(guest<16>, dim3(1), dim3(g_threads), (size_t)g_lds_kib * 1024, s1, tg, sink);
            else hipLaunchKernelGGL(guest<96>, dim3(1), dim3(g_threads), (size_t)g_lds_kib * 1024, s1, tg, sink);
        }
        if (hog_kind == 0) hipLaunchKernelGGL(hog, dim3(hog_blocks), dim3(256), (size_t)hog_lds_kib * 1024, s0, th, iters, sink, 256);
        else if (hog_kind == 1) hipLaunchKernelGGL(hog_regs<80>, dim3(hog_blocks), dim3(256), 0, s0, th, iters / 160, sink);
        else if (hog_kind == 2) hipLaunchKernelGGL(hog_regs<120>, dim3(hog_blocks), dim3(256), 0, s0, th, iters / 240, sink);
        else if (hog_kind == 10) hipLaunchKernelGGL(hog_mfma<0>, dim3(hog_blocks), dim3(256), 0, s0, th, 1200, sink);
        else if (hog_kind == 11) hipLaunchKernelGGL(hog_mfma<1>, dim3(hog_blocks), dim3(256), 0, s0, th, 1200, sink);
        else hipLaunchKernelGGL(hog_mfma<2>, dim3(hog_blocks), dim3(256), 0, s0, th, 1200, sink);
        if (!order_guest_first) {
            if (g_regs <= 16) hipLaunchKernelGGL(guest<16>, dim3(1), dim3(g_threads), (size_t)g_lds_kib * 1024, s1, tg, sink);
            else hipLaunchKernelGGL(guest<96>, dim3(1), dim3(g_threads), (size_t)g_lds_kib * 1024, s1, tg, sink);
        }
        CK(hipDeviceSynchronize());
        std::vector<long long> h(2 * hog_blocks); long long g[2];
        CK(hipMemcpy(h.data(), th, sizeof(long long) * 2 * hog_blocks, hipMemcpyDeviceToHost)); CK(hipMemcpy(g, tg, sizeof g, hipMemcpyDeviceToHost));
        long long h0 = h[0], h1 = 0, hfirst_end = h[hog_blocks];
        for (int i = 0; i < hog_blocks; ++i) { if (h[i] < h0) h0 = h[i]; if (h[hog_blocks + i] > h1) h1 = h[hog_blocks + i]; if (h[hog_blocks + i] < hfirst_end) hfirst_end = h[hog_blocks + i]; }
        const double us = 1.0 / 100.0;   // wall_clock64: 100 MHz
        printf("hog %3d blocks x %3d KiB (first end %.0f us, last end %.0f us) | guest %4d thr, %2d regs, %3d KiB, %s: first instruction at %+8.1f us, body %.1f us\n", hog_blocks, hog_lds_kib,
               (hfirst_end - h0) * us, (h1 - h0) * us, g_threads, g_regs, g_lds_kib, order_guest_first ? "launched first" : "launched second", (g[0] - h0) * us, (g[1] - g[0]) * us);
        return 0;
    };
    for (hog_kind = 10; hog_kind <= 12; ++hog_kind)
        for (guest_prio = 0; guest_prio <= 1; ++guest_prio) {
            printf("MFMA hog, yield form %d, guest %s\n", hog_kind - 10, guest_prio ? "s_setprio 3" : "default priority");
            if (run(ncu, 64, 512, 96, 70, false)) return 1;
            if (run(ncu, 64, 512, 16, 1, false)) return 1;
            if (run(ncu, 64, 64, 16, 1, false)) return 1;
        }
    for (hog_kind = 10; hog_kind <= 12; hog_kind += 2)
        for (guest_prio = 0; guest_prio <= 1; ++guest_prio) {
            printf("MFMA hog, yield form %d, guest LAUNCHED FIRST, %s\n", hog_kind - 10, guest_prio ? "s_setprio 3" : "default priority");
            if (run(ncu, 64, 512, 96, 70, true)) return 1;
            if (run(ncu, 64, 512, 16, 1, true)) return 1;
            if (run(2 * ncu - 8, 64, 512, 96, 70, true)) return 1;
        }
    for (guest_prio = 1, hog_kind = 1; hog_kind <= 1; ++hog_kind) {
        printf("VALU hog, guest s_setprio 3\n");
        if (run(ncu, 64, 512, 16, 1, false)) return 1;
    }
    guest_prio = 0;
    for (hog_kind = 1; hog_kind <= 0; ++hog_kind) {
        printf("hog kind %d (register-heavy, 64 KiB static LDS)\n", hog_kind);
        if (run(ncu, 64, 512, 96, 70, false)) return 1;
        if (run(ncu, 64, 512, 16, 70, false)) return 1;
        if (run(ncu, 64, 512, 16, 1, false)) return 1;
        if (run(ncu, 64, 64, 16, 1, false)) return 1;
        if (run(2 * ncu - 8, 64, 512, 96, 70, false)) return 1;
    }
    hog_kind = 0;
    for (int rep = 0; rep < 0; ++rep) {
        if (run(ncu, 64, 512, 96, 70, false)) return 1;        // one hog block per CU (the unsplit product): the potrf's shape
        if (run(ncu, 64, 512, 16, 70, false)) return 1;
        if (run(ncu, 64, 512, 16, 1, false)) return 1;
        if (run(ncu, 64, 256, 16, 1, false)) return 1;
        if (run(ncu, 64, 64, 16, 1, false)) return 1;
        if (run(2 * ncu - 8, 64, 512, 96, 70, false)) return 1;  // the short 2-per-CU grid
        if (run(2 * ncu - 8, 64, 256, 16, 1, false)) return 1;
        if (run(ncu, 64, 512, 96, 70, true)) return 1;         // guest first
        if (run(ncu, 64, 512, 96, 100, true)) return 1;        // guest first with LDS that excludes a hog block
        if (run(ncu, 1, 512, 96, 70, false)) return 1;         // hog without LDS to speak of
    }
    return 0;
}
