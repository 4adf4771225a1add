// What does synchronisation INSIDE a persistent grid cost on this box, with the barrier built the way the hardware guide prescribes
// (MI355X_MICROARCH.md, price list rows barrier-xcd / barrier-counter / boundary) -- the question behind the persistent 2-D form of the
// small-k MultUpdate iteration, whose round-5 rejection rested on a single-counter barrier polled with s_sleep
// by groups that straddled XCDs (scripts/kbench/group_barrier_probe.hip: 14 us per 16-party barrier).
//
// 256 workgroups, one per CU.  Block b is assumed to run on XCD b % 8 (observed placement, used for SPEED only: every protocol below
// counts arrivals, none depends on where a block runs).
//   mode 0  grid barrier, ONE counter (barrier-counter row): arrive = lane-0 release fence + relaxed agent add, poll = sc1 load + s_sleep
//   mode 1  grid barrier, XCD-hierarchical (barrier-xcd row): per-XCC counter; the LAST arriver of an XCC (its leader for this
//           generation) adds to the top counter, polls it, acquires, and publishes the generation in a per-XCC word that the 31 other
//           blocks of the XCC poll (sc1 loads served by that XCD's L2); everybody acquires
//   mode 2  16-party group barriers, groups INSIDE one XCD (blocks b with equal b % 8, two groups of 16 per XCD), one counter per group
//   mode 3  the same 16-party groups STRADDLING the XCDs (16 consecutive block ids: two blocks per XCD) -- round 5's placement
//   mode 4  the synchronisation skeleton of one persistent 2-D-blocked iteration: 4 in-XCD group barriers + 1 hierarchical grid barrier
//   mode 5  mode 4 with the exchange traffic of the blocked iteration (64 KB partial written, 16 x 4 KB slices read, 4 KB written,
//           64 KB read, per side)
// and, for comparison, mode 6: the SAME number of synchronisation points as dependent launches of an (almost) empty 256-block kernel on
// one stream (the boundary row: what the shipped four-launch small-k iteration pays per synchronisation point).
// Every spin is bounded (2^22 polls, then an abort word every other spin sees): a mis-launch cannot hang the GPU.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/kbench/xcd_barrier_probe.hip -o /tmp/xcd_barrier_probe && /tmp/xcd_barrier_probe 500
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int NB = 256, NX = 8, PER = NB / NX, G = 16, STRIDE = 32;   // STRIDE ints = 128 bytes between synchronisation words
struct Sync { int *xcc, *gen, *top, *one, *grp, *abort; };

__device__ __forceinline__ int ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void spin_until(const int *p, int target, int *abort) {
    int spins = 0;
    while (ld(p) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22) || ld(abort)) { __hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
}
// one counter, n parties; gen = 1, 2, ...
__device__ __forceinline__ void counter_barrier(int *ctr, int n, int gen, int *abort) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        spin_until(ctr, n * gen, abort);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
// XCD-hierarchical grid barrier
__device__ __forceinline__ void xcd_barrier(const Sync &s, int gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const int x = blockIdx.x % NX;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const int old = __hip_atomic_fetch_add(s.xcc + x * STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == PER * gen - 1) {   // the XCC's last arriver: its leader for this generation
            __hip_atomic_fetch_add(s.top, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            spin_until(s.top, NX * gen, s.abort);
            __hip_atomic_store(s.gen + x * STRIDE, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            spin_until(s.gen + x * STRIDE, gen, s.abort);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
template <int MODE>
__global__ __launch_bounds__(512) void probe(Sync s, float *part, float *slices, float *sink, int iters) {
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    __shared__ float blk[64 * 256];
    float acc = 0.f;
    // in-XCD groups: blocks with equal b % 8, halves by (b / 8) / 16;  straddling groups: 16 consecutive ids
    const int gi = (MODE == 3) ? b / G : (b % NX) * 2 + (b / NX) / G, me = (MODE == 3) ? b % G : (b / NX) % G;
    int ggen = 0, xgen = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) counter_barrier(s.one, NB, ++xgen, s.abort);
        else if (MODE == 1) xcd_barrier(s, ++xgen);
        else if (MODE == 2 || MODE == 3) counter_barrier(s.grp + gi * STRIDE, G, ++ggen, s.abort);
        else {
            for (int side = 0; side < 2; ++side) {
                float *P = part + ((size_t)side * NB + (size_t)gi * G) * 16384, *S = slices + ((size_t)side * (NB / G) + gi) * 16384;
                if (MODE == 5) {
                    float4 *dst = reinterpret_cast<float4 *>(P + (size_t)me * 16384);
                    for (int e = tid; e < 4096; e += nt) dst[e] = float4{acc, (float)e, (float)it, 1.f};
                }
                counter_barrier(s.grp + gi * STRIDE, G, ++ggen, s.abort);
                if (MODE == 5) {
                    float sum = 0.f;
                    for (int e0 = tid; e0 < 4096; e0 += 8 * nt) {
                        float4 v[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) { const int e = (e0 + q * nt) & 4095; v[q] = reinterpret_cast<const float4 *>(P + (size_t)(e >> 8) * 16384 + (size_t)me * 1024)[e & 255]; }
#pragma unroll
                        for (int q = 0; q < 8; ++q) sum += v[q].x + v[q].y + v[q].z + v[q].w;
                    }
                    acc += sum;
                    if (tid < 256) reinterpret_cast<float4 *>(S + (size_t)me * 1024)[tid] = float4{sum, acc, 0.f, 0.f};
                }
                if (side == 0) counter_barrier(s.grp + gi * STRIDE, G, ++ggen, s.abort);
                else xcd_barrier(s, ++xgen);
                if (MODE == 5) {
                    for (int e = tid; e < 4096; e += nt) reinterpret_cast<float4 *>(blk)[e] = reinterpret_cast<const float4 *>(S)[e];
                    __syncthreads();
                    acc += blk[(tid * 33) & 16383];
                }
            }
        }
    }
    sink[b * 512 + tid] = acc;
}
__global__ void tiny(float *sink) { if (threadIdx.x == 0) sink[blockIdx.x] += 1.f; }

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 500;
    Sync s; int *w;
    const int nwords = (2 * NX + 2 + NB / G + 1) * STRIDE;
    CK(hipMalloc(&w, nwords * sizeof(int)));
    s.xcc = w; s.gen = w + NX * STRIDE; s.top = w + 2 * NX * STRIDE; s.one = s.top + STRIDE; s.grp = s.one + STRIDE; s.abort = s.grp + (NB / G) * STRIDE;
    float *part, *slices, *sink;
    CK(hipMalloc(&part, (size_t)2 * NB * 16384 * 4)); CK(hipMalloc(&slices, (size_t)2 * (NB / G) * 16384 * 4)); CK(hipMalloc(&sink, NB * 512 * 4));
    CK(hipMemset(sink, 0, NB * 512 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *what[] = {"grid barrier, one counter (barrier-counter)", "grid barrier, XCD-hierarchical (barrier-xcd)", "16-party group barrier, group inside one XCD",
                          "16-party group barrier, group straddling the XCDs", "2-D iteration skeleton: 4 in-XCD group barriers + 1 XCD-hierarchical grid barrier",
                          "the same skeleton with the blocked iteration's exchange traffic", "dependent launches of an empty 256-block kernel (boundary)"};
    for (int nt : {256, 512})
        for (int mode = 0; mode < 7; ++mode)
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(w, 0, nwords * sizeof(int)));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                switch (mode) {
                    case 0: hipLaunchKernelGGL(probe<0>, dim3(NB), dim3(nt), 0, 0, s, part, slices, sink, iters); break;
                    case 1: hipLaunchKernelGGL(probe<1>, dim3(NB), dim3(nt), 0, 0, s, part, slices, sink, iters); break;
                    case 2: hipLaunchKernelGGL(probe<2>, dim3(NB), dim3(nt), 0, 0, s, part, slices, sink, iters); break;
                    case 3: hipLaunchKernelGGL(probe<3>, dim3(NB), dim3(nt), 0, 0, s, part, slices, sink, iters); break;
                    case 4: hipLaunchKernelGGL(probe<4>, dim3(NB), dim3(nt), 0, 0, s, part, slices, sink, iters); break;
                    case 5: hipLaunchKernelGGL(probe<5>, dim3(NB), dim3(nt), 0, 0, s, part, slices, sink, iters); break;
                    default: for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(tiny, dim3(NB), dim3(nt), 0, 0, sink); break;
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                int ab; CK(hipMemcpy(&ab, s.abort, 4, hipMemcpyDeviceToHost));
                if (rep > 0) printf("%3d threads  mode %d  %-88s %8.2f us per %s  abort=%d\n", nt, mode, what[mode], ms * 1e3 / iters, mode >= 4 && mode < 6 ? "iteration" : "barrier  ", ab);
            }
    return 0;
}
