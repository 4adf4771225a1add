// What does a persistent 2-D grid pay for its synchronisation?  256 workgroups (16 x 16, one per CU, 512 threads) run `iters` rounds of
//   [write 64 KB partial -> 16-party COLUMN-group barrier -> read 16 x 4 KB slices -> write 4 KB -> column-group barrier -> read 64 KB]
//   [the same over the ROW group]  -> two-level grid barrier (row-group counters, then one global counter)
// i.e. the exchange skeleton of a 2-D-blocked small-k MultUpdate iteration without any arithmetic.
// Barriers: monotone counters in global memory, one per group, 128 bytes apart; arrive = release fence + relaxed atomic add (agent
// scope), wait = spin on a relaxed load with s_sleep, acquire fence after; every spin is bounded (a wave that spins 2^22 times gives up
// and raises `abort`, which every other spin sees): a mis-launch cannot hang the GPU.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/kbench/group_barrier_probe.hip -o /tmp/group_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int G = 16, NT = 512, STRIDE = 32;   // ints between counters
struct Sync { int *col, *row, *rowg, *all, *abort; };
__device__ __forceinline__ bool wait_for(int *ctr, int target, int *abort) {
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22) || __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { __hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
}
__device__ __forceinline__ void group_barrier(int *ctr, int gen, int *abort) {   // gen = 1, 2, ...: the counter reaches G * gen
    __syncthreads();
    if (threadIdx.x == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    wait_for(ctr, G * gen, abort);
}
// two levels: the last arriver of a row group bumps the global counter; everybody waits for the global counter
__device__ __forceinline__ void grid_barrier(int *rowctr, int *all, int gen, int *abort) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const int old = __hip_atomic_fetch_add(rowctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == G * gen - 1) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent"); __hip_atomic_fetch_add(all, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
    wait_for(all, G * gen, abort);
}
template <int MODE>   // 0: barriers only; 1: + the exchange traffic
__global__ __launch_bounds__(NT) void probe(Sync s, float *part, float *slices, float *sink, int iters) {
    const int bi = blockIdx.x / G, bj = blockIdx.x % G, tid = threadIdx.x;
    __shared__ float blk[64 * 256];
    float acc = 0.f;
    int cgen = 0, rgen = 0, ggen = 0;
    for (int it = 0; it < iters; ++it) {
        for (int side = 0; side < 2; ++side) {
            int *ctr = side == 0 ? s.col + bj * STRIDE : s.row + bi * STRIDE;
            int &gen = side == 0 ? cgen : rgen;
            const int grp = side == 0 ? bj : bi, me = side == 0 ? bi : bj;   // group id, index inside the group
            float *P = part + ((size_t)side * 256 + (size_t)grp * G) * 16384, *S = slices + ((size_t)side * G + grp) * 16384;
            if (MODE == 1) {   // my partial: 64 x 256 floats
                float4 *dst = reinterpret_cast<float4 *>(P + (size_t)me * 16384);
                for (int e = tid; e < 4096; e += NT) dst[e] = float4{acc, (float)e, (float)it, 1.f};
            }
            group_barrier(ctr, ++gen, s.abort);
            if (MODE == 1) {   // my slice of all 16 partials: 16 x 1024 floats
                float4 v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { const int e = tid + NT * q; v[q] = reinterpret_cast<const float4 *>(P + (size_t)(e >> 8) * 16384 + (size_t)me * 1024)[e & 255]; }
                float sum = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) sum += v[q].x + v[q].y + v[q].z + v[q].w;
                acc += sum;
                if (tid < 256) reinterpret_cast<float4 *>(S + (size_t)me * 1024)[tid] = float4{sum, acc, 0.f, 0.f};
            }
            if (side == 0) group_barrier(ctr, ++gen, s.abort);
            else grid_barrier(s.rowg + bi * STRIDE, s.all, ++ggen, s.abort);
            if (MODE == 1) {   // the updated block of the group: 64 KB into LDS
                for (int e = tid; e < 4096; e += NT) reinterpret_cast<float4 *>(blk)[e] = reinterpret_cast<const float4 *>(S)[e];
                __syncthreads();
                acc += blk[(tid * 33) & 16383];
            }
        }
    }
    sink[blockIdx.x * NT + tid] = acc;
}
int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    Sync s; int *ctrs;
    CK(hipMalloc(&ctrs, (3 * G + 2) * STRIDE * sizeof(int)));
    s.col = ctrs; s.row = ctrs + G * STRIDE; s.rowg = ctrs + 2 * G * STRIDE; s.all = ctrs + 3 * G * STRIDE; s.abort = s.all + STRIDE;
    float *part, *slices, *sink;
    CK(hipMalloc(&part, (size_t)2 * 256 * 16384 * 4)); CK(hipMalloc(&slices, (size_t)2 * G * 16384 * 4)); CK(hipMalloc(&sink, 256 * NT * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(ctrs, 0, (3 * G + 2) * STRIDE * sizeof(int)));
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(NT), 0, 0, s, part, slices, sink, iters);
            else hipLaunchKernelGGL(probe<1>, dim3(256), dim3(NT), 0, 0, s, part, slices, sink, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            int ab; CK(hipMemcpy(&ab, s.abort, 4, hipMemcpyDeviceToHost));
            printf("mode %d (%s): %d iterations, %.2f us per iteration (4 group barriers + 1 grid barrier%s), abort=%d\n", mode, mode ? "barriers + exchange traffic" : "barriers only", iters,
                   ms * 1e3 / iters, mode ? ", 2 x (64 KB written, 64 + 4 KB... read/written, 64 KB read)" : "", ab);
        }
    return 0;
}
