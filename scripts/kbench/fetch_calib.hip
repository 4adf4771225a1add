// fetch_calib.hip -- what do rocprofv3's FETCH_SIZE / WRITE_SIZE report per byte for the access widths the kernels of this library use?
// (MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of a wide coalesced read on gfx950; "other access widths and WRITE_SIZE are
// uncalibrated: calibrate on a known byte count in your own access pattern".)  Each kernel moves exactly BYTES = 1 GiB once:
//   read16   global_load_dwordx4 per lane (the GEMM operand staging)          read4_buf  4-byte buffer loads, lanes on consecutive floats
//   write4_buf  4-byte buffer stores (the epilogues: X read / Q written by the ratio pass)   write16  16-byte stores
// run:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- ./fetch_calib   (and once more with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void read16(const v4f *p, size_t n, float *sink) {
    v4f acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = acc[0];
}
__global__ void read4_buf(const float *p, size_t n, float *sink) {
    float acc = 0.f;
    // a wave covers 64 consecutive floats per load, the block walks 1 MiB chunks: offsets stay inside 32 bits per descriptor
    const size_t chunk = (size_t)1 << 18;   // floats per descriptor
    for (size_t c = blockIdx.x; c * chunk < n; c += gridDim.x) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(p + c * chunk), 0, -1, 0x00020000);
        for (unsigned i = threadIdx.x; i < chunk; i += blockDim.x) acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(i * 4), 0, 0));
    }
    if (acc == 12345.678f) *sink = acc;
}
__global__ void write4_buf(float *p, size_t n) {
    const size_t chunk = (size_t)1 << 18;
    for (size_t c = blockIdx.x; c * chunk < n; c += gridDim.x) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(p + c * chunk), 0, -1, 0x00020000);
        for (unsigned i = threadIdx.x; i < chunk; i += blockDim.x) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)i), rs, (int)(i * 4), 0, 0);
    }
}
__global__ void write16(v4f *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v4f{1.f, 2.f, 3.f, (float)i};
}
int main() {
    const size_t BYTES = (size_t)1 << 30;
    float *a, *sink;
    CK(hipMalloc(&a, BYTES)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 0, BYTES));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read16, dim3(2048), dim3(256), 0, 0, (const v4f *)a, BYTES / 16, sink);
        hipLaunchKernelGGL(read4_buf, dim3(1024), dim3(256), 0, 0, a, BYTES / 4, sink);
        hipLaunchKernelGGL(write4_buf, dim3(1024), dim3(256), 0, 0, a, BYTES / 4);
        hipLaunchKernelGGL(write16, dim3(2048), dim3(256), 0, 0, (v4f *)a, BYTES / 16);
    }
    CK(hipDeviceSynchronize());
    printf("moved %zu bytes per kernel launch\n", BYTES);
    return 0;
}
