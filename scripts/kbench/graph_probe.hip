// Does a hipGraph replay shorten a chain of short dependent kernels on this platform?  Chain: 4 kernels (2 x ~10 us of work on all
// CUs, 2 x tiny) per "iteration", 8 iterations per graph, against the same chain enqueued launch by launch on a stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void work(float *x, int spin) {
    float a = x[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    x[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main() {
    float *d; CK(hipMalloc(&d, 256 * 512 * sizeof(float)));
    CK(hipMemset(d, 0, 256 * 512 * sizeof(float)));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto chain = [&](hipStream_t st) {
        hipLaunchKernelGGL(work, dim3(256), dim3(512), 0, st, d, 2000);
        hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, st, d, 50);
        hipLaunchKernelGGL(work, dim3(256), dim3(512), 0, st, d, 2000);
        hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, st, d, 50);
    };
    const int iters = 2000;
    for (int i = 0; i < 50; ++i) chain(s);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) chain(s);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("stream launches : %.2f us per 4-kernel iteration\n", ms * 1000 / iters);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 8; ++i) chain(s);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters / 8; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("graph (8 iters) : %.2f us per 4-kernel iteration\n", ms * 1000 / iters);
    // single kernel duration for reference
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(work, dim3(256), dim3(512), 0, s, d, 2000);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("big kernel alone, back to back: %.2f us each\n", ms * 1000 / 200);
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s, d, 50);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("tiny kernel alone, back to back: %.2f us each\n", ms * 1000 / 200);
    return 0;
}
