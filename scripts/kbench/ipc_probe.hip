// ipc_probe.hip -- what does the one-process-per-GPU peer-to-peer exchange have to work with on this stack?
//
// Forks R processes BEFORE any HIP call (each opens device `dev`, default 0 -- several processes on ONE GPU, which is all the
// 1-GPU box offers), every process allocates a window, exports it with hipIpcGetMemHandle, imports the others', then:
//   1. which allocation kinds can be exported / imported (hipMalloc, fine-grained, uncached)
//   2. flag ping-pong between rank 0 and rank 1 (kernel A writes a flag into the peer's window, kernel B spins on it): round trip
//   3. tiny all-reduce (3 doubles) written as {value, seq} granules into every peer's inbox, spun on inside ONE kernel: latency
//   4. push all-gather: every rank stores 2 MiB into every peer's slot, then a flag; consumers wait and checksum
// Build: hipcc --offload-arch=gfx950 -O3 ipc_probe.hip -o ipc_probe ; run: ./ipc_probe [ranks=4] [dev=0]
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>

#define CK(x)                                                                                         \
    do {                                                                                              \
        hipError_t _e = (x);                                                                          \
        if (_e != hipSuccess) {                                                                       \
            std::fprintf(stderr, "[rank %d] %s:%d %s -> %s\n", g_rank, __FILE__, __LINE__, #x, hipGetErrorString(_e)); \
            std::exit(3);                                                                             \
        }                                                                                             \
    } while (0)

static int g_rank = -1;
constexpr int MAXR = 16;

struct Shared {
    std::atomic<int> arrive;
    std::atomic<int> gen;
    hipIpcMemHandle_t h[3][MAXR];
    int ok[3][MAXR];
    double result[MAXR][8];
};

static void host_barrier(Shared *s, int n) {
    const int g = s->gen.load();
    if (s->arrive.fetch_add(1) + 1 == n) {
        s->arrive.store(0);
        s->gen.fetch_add(1);
    } else {
        while (s->gen.load() == g) usleep(50);
    }
}

// ---- device side ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void sys_store_u32(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ unsigned sys_load_u32(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

__global__ void set_flag_kernel(unsigned *flag, unsigned v) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    sys_store_u32(flag, v);
}
__global__ void wait_flag_kernel(const unsigned *flag, unsigned v, unsigned *timeout) {
    if (*timeout) return;
    long long spins = 0;
    while (sys_load_u32(flag) < v) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > 4000000ll) { *timeout = 1; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

struct Peers { void *p[MAXR]; };

// tiny all-reduce: granule = {double value, unsigned long long seq}: 16 bytes written by ONE 16-byte store
typedef unsigned long long u64;
struct __attribute__((aligned(16))) Gran { double v; u64 seq; };
__global__ void tiny_allreduce_kernel(Peers win, int rank, int n, const double *in, double *out, int nval, u64 seq, unsigned *timeout) {
    // inbox layout in every window: Gran inbox[2][MAXR][8]  (double-buffered by seq parity)
    const int t = threadIdx.x;
    if (*timeout) return;
    const int par = (int)(seq & 1);
    if (t < n * nval) {
        const int q = t / nval, j = t % nval;
        Gran *dst = reinterpret_cast<Gran *>(win.p[q]) + ((par * MAXR + rank) * 8 + j);
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        Gran g{in[j], seq};
        v4u w = __builtin_bit_cast(v4u, g);
        // one 16-byte system-scope (sc0 sc1) store: value and tag land together
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(w) : "memory");
    }
    __syncthreads();
    if (t < nval) {
        double s = 0.0;
        for (int q = 0; q < n; ++q) {
            const Gran *src = reinterpret_cast<const Gran *>(win.p[rank]) + ((par * MAXR + q) * 8 + t);
            long long spins = 0;
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            v4u w;
            while (true) {
                asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(w) : "v"(src) : "memory");
                Gran g = __builtin_bit_cast(Gran, w);
                if (g.seq == seq) { s += g.v; break; }
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000ll) { *timeout = 1; break; }
            }
        }
        out[t] = s;
    }
}

// push: every rank stores `count` floats into slot `rank` of every peer's data area (sc0 sc1 stores)
__global__ void push_kernel(Peers data, int rank, int n, const float *src, size_t count) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count / 4; i += (size_t)gridDim.x * blockDim.x) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f v = reinterpret_cast<const v4f *>(src)[i];
        for (int q = 0; q < n; ++q) {
            v4f *dst = reinterpret_cast<v4f *>(reinterpret_cast<float *>(data.p[q]) + (size_t)rank * count) + i;
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
        }
    }
}
__global__ void signal_all_kernel(Peers flags, int rank, int n, unsigned v) {
    const int q = threadIdx.x;
    if (q < n) sys_store_u32(reinterpret_cast<unsigned *>(flags.p[q]) + 64 + rank, v);
}
__global__ void wait_sum_kernel(const unsigned *myflags, int n, unsigned v, const float *slots, size_t count, double *out, unsigned *timeout) {
    __shared__ double sm[4];
    if (*timeout) return;
    if (threadIdx.x < n) {
        long long spins = 0;
        while (sys_load_u32(myflags + 64 + threadIdx.x) < v) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > 4000000ll) { *timeout = 1; break; }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count * n; i += (size_t)gridDim.x * blockDim.x) s += slots[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, sm[0] + sm[1] + sm[2] + sm[3]);
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int run_rank(Shared *sh, int rank, int n, int dev) {
    g_rank = rank;
    setvbuf(stdout, nullptr, _IONBF, 0);
    CK(hipSetDevice(dev));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t FLAG_BYTES = 1 << 16;                 // flags + inbox
    const size_t SLOT_FLOATS = (size_t)512 * 1024;     // 2 MiB per (rank, slot)
    const size_t DATA_BYTES = SLOT_FLOATS * sizeof(float) * MAXR;
    void *win[3] = {nullptr, nullptr, nullptr};
    const char *kind[3] = {"hipMalloc", "finegrained", "uncached"};
    // kind 0: plain hipMalloc (data), 1: fine-grained, 2: uncached
    CK(hipMalloc(&win[0], DATA_BYTES));
    hipError_t e1 = hipExtMallocWithFlags(&win[1], FLAG_BYTES, hipDeviceMallocFinegrained);
    hipError_t e2 = hipExtMallocWithFlags(&win[2], FLAG_BYTES, hipDeviceMallocUncached);
    if (e1 != hipSuccess) { std::fprintf(stderr, "[rank %d] finegrained alloc: %s\n", rank, hipGetErrorString(e1)); win[1] = nullptr; (void)hipGetLastError(); }
    if (e2 != hipSuccess) { std::fprintf(stderr, "[rank %d] uncached alloc: %s\n", rank, hipGetErrorString(e2)); win[2] = nullptr; (void)hipGetLastError(); }
    for (int k = 0; k < 3; ++k) {
        sh->ok[k][rank] = 0;
        if (!win[k]) continue;
        CK(hipMemset(win[k], 0, k == 0 ? DATA_BYTES : FLAG_BYTES));
        hipError_t e = hipIpcGetMemHandle(&sh->h[k][rank], win[k]);
        if (e != hipSuccess) { std::fprintf(stderr, "[rank %d] hipIpcGetMemHandle(%s): %s\n", rank, kind[k], hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        sh->ok[k][rank] = 1;
    }
    CK(hipDeviceSynchronize());
    host_barrier(sh, n);
    if (rank == 0) std::printf("[ipc] %d processes: windows allocated and exported\n", n);
    Peers peers[3];
    for (int k = 0; k < 3; ++k) {
        for (int q = 0; q < MAXR; ++q) peers[k].p[q] = nullptr;
        for (int q = 0; q < n; ++q) {
            if (q == rank) { peers[k].p[q] = win[k]; continue; }
            if (!sh->ok[k][q]) continue;
            void *ptr = nullptr;
            hipError_t e = hipIpcOpenMemHandle(&ptr, sh->h[k][q], hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) { std::fprintf(stderr, "[rank %d] hipIpcOpenMemHandle(%s of rank %d): %s\n", rank, kind[k], q, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
            peers[k].p[q] = ptr;
        }
        bool all = true;
        for (int q = 0; q < n; ++q) all = all && peers[k].p[q] != nullptr;
        if (rank == 0) std::printf("[ipc] %-12s export+import across %d processes on device %d: %s\n", kind[k], n, dev, all ? "OK" : "FAILED");
        if (!all) for (int q = 0; q < MAXR; ++q) peers[k].p[q] = nullptr;
    }
    host_barrier(sh, n);
    // flags live in the first kind that works among (uncached, finegrained, hipMalloc)
    int fk = peers[2].p[0] ? 2 : (peers[1].p[0] ? 1 : 0);
    if (!peers[fk].p[0]) { std::fprintf(stderr, "[rank %d] no shareable allocation kind\n", rank); return 4; }
    if (rank == 0) std::printf("[ipc] flags in %s memory\n", kind[fk]);
    unsigned *timeout;
    CK(hipMalloc(reinterpret_cast<void **>(&timeout), 4));
    CK(hipMemset(timeout, 0, 4));
    double *dio;
    CK(hipMalloc(reinterpret_cast<void **>(&dio), 64 * sizeof(double)));
    CK(hipMemset(dio, 0, 64 * sizeof(double)));

    // 2. flag ping-pong between rank 0 and 1
    if (n >= 2 && rank < 2) {
        const int peer = rank ^ 1;
        unsigned *mine = reinterpret_cast<unsigned *>(peers[fk].p[rank]), *theirs = reinterpret_cast<unsigned *>(peers[fk].p[peer]);
        const int ITER = 200;
        CK(hipDeviceSynchronize());
    }
    host_barrier(sh, n);
    if (n >= 2 && rank < 2) {
        const int peer = rank ^ 1;
        unsigned *mine = reinterpret_cast<unsigned *>(peers[fk].p[rank]), *theirs = reinterpret_cast<unsigned *>(peers[fk].p[peer]);
        const int ITER = 200;
        const double t0 = now_us();
        for (int i = 1; i <= ITER; ++i) {
            if (rank == 0) {
                hipLaunchKernelGGL(set_flag_kernel, dim3(1), dim3(1), 0, st, theirs, (unsigned)i);
                hipLaunchKernelGGL(wait_flag_kernel, dim3(1), dim3(1), 0, st, mine, (unsigned)i, timeout);
            } else {
                hipLaunchKernelGGL(wait_flag_kernel, dim3(1), dim3(1), 0, st, mine, (unsigned)i, timeout);
                hipLaunchKernelGGL(set_flag_kernel, dim3(1), dim3(1), 0, st, theirs, (unsigned)i);
            }
        }
        CK(hipStreamSynchronize(st));
        const double t1 = now_us();
        unsigned to = 0;
        CK(hipMemcpy(&to, timeout, 4, hipMemcpyDeviceToHost));
        if (rank == 0) std::printf("[ipc] flag ping-pong (2 launches per hop): %.2f us per round trip, timeout=%u\n", (t1 - t0) / ITER, to);
    }
    host_barrier(sh, n);

    // 3. tiny all-reduce inside one kernel
    {
        double h_in[3] = {1.0 + rank, 0.5 * (rank + 1), -2.0 * rank};
        CK(hipMemcpy(dio, h_in, sizeof h_in, hipMemcpyHostToDevice));
        const int ITER = 500;
        host_barrier(sh, n);
        const double t0 = now_us();
        for (int i = 1; i <= ITER; ++i)
            hipLaunchKernelGGL(tiny_allreduce_kernel, dim3(1), dim3(64), 0, st, peers[fk], rank, n, dio, dio + 8, 3, (u64)i, timeout);
        CK(hipStreamSynchronize(st));
        const double t1 = now_us();
        double h_out[3];
        unsigned to = 0;
        CK(hipMemcpy(h_out, dio + 8, sizeof h_out, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&to, timeout, 4, hipMemcpyDeviceToHost));
        double e0 = 0, e1 = 0, e2 = 0;
        for (int q = 0; q < n; ++q) { e0 += 1.0 + q; e1 += 0.5 * (q + 1); e2 += -2.0 * q; }
        const bool ok = h_out[0] == e0 && h_out[1] == e1 && h_out[2] == e2;
        if (rank == 0) std::printf("[ipc] in-kernel all-reduce of 3 doubles over %d processes: %.2f us per call, %s, timeout=%u\n", n, (t1 - t0) / ITER, ok ? "correct" : "WRONG", to);
        if (!ok) std::fprintf(stderr, "[rank %d] all-reduce got %g %g %g want %g %g %g\n", rank, h_out[0], h_out[1], h_out[2], e0, e1, e2);
    }
    host_barrier(sh, n);

    // 4. push all-gather of 2 MiB per rank + flags + consumer checksum
    {
        float *src;
        CK(hipMalloc(reinterpret_cast<void **>(&src), SLOT_FLOATS * sizeof(float)));
        float *hsrc = (float *)std::malloc(SLOT_FLOATS * sizeof(float));
        for (size_t i = 0; i < SLOT_FLOATS; ++i) hsrc[i] = (float)((i * 7 + rank * 13) % 97) * 0.25f;
        CK(hipMemcpy(src, hsrc, SLOT_FLOATS * sizeof(float), hipMemcpyHostToDevice));
        const int ITER = 50;
        host_barrier(sh, n);
        double got = 0.0;
        const double t0 = now_us();
        for (int i = 1; i <= ITER; ++i) {
            hipLaunchKernelGGL(push_kernel, dim3(512), dim3(256), 0, st, peers[0], rank, n, src, SLOT_FLOATS);
            hipLaunchKernelGGL(signal_all_kernel, dim3(1), dim3(64), 0, st, peers[fk], rank, n, (unsigned)i);
            CK(hipMemsetAsync(dio + 16, 0, 8, st));
            hipLaunchKernelGGL(wait_sum_kernel, dim3(512), dim3(256), 0, st, reinterpret_cast<const unsigned *>(peers[fk].p[rank]), n, (unsigned)i,
                               reinterpret_cast<const float *>(peers[0].p[rank]), SLOT_FLOATS, dio + 16, timeout);
        }
        CK(hipStreamSynchronize(st));
        const double t1 = now_us();
        CK(hipMemcpy(&got, dio + 16, 8, hipMemcpyDeviceToHost));
        double want = 0.0;
        for (int q = 0; q < n; ++q)
            for (size_t i = 0; i < SLOT_FLOATS; ++i) want += (double)((float)((i * 7 + q * 13) % 97) * 0.25f);
        unsigned to = 0;
        CK(hipMemcpy(&to, timeout, 4, hipMemcpyDeviceToHost));
        if (rank == 0)
            std::printf("[ipc] push all-gather 2 MiB x %d ranks + flag + checksum: %.1f us per round, checksum %s (%.1f vs %.1f), timeout=%u\n", n,
                        (t1 - t0) / ITER, std::fabs(got - want) < 1e-6 * want ? "OK" : "WRONG", got, want, to);
        std::free(hsrc);
    }
    host_barrier(sh, n);
    for (int k = 0; k < 3; ++k)
        for (int q = 0; q < n; ++q)
            if (q != rank && peers[k].p[q]) (void)hipIpcCloseMemHandle(peers[k].p[q]);
    host_barrier(sh, n);
    return 0;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 4;
    const int dev = argc > 2 ? std::atoi(argv[2]) : 0;
    if (n < 1 || n > MAXR) return 2;
    Shared *sh = reinterpret_cast<Shared *>(mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0));
    if (sh == MAP_FAILED) return 2;
    new (sh) Shared();
    sh->arrive.store(0);
    sh->gen.store(0);
    pid_t pids[MAXR];
    for (int r = 0; r < n; ++r) {
        pids[r] = fork();
        if (pids[r] == 0) _exit(run_rank(sh, r, n, dev));
    }
    int bad = 0;
    for (int r = 0; r < n; ++r) {
        int stt = 0;
        waitpid(pids[r], &stt, 0);
        if (!WIFEXITED(stt) || WEXITSTATUS(stt) != 0) { std::fprintf(stderr, "rank %d exited abnormally (%d)\n", r, stt); bad = 1; }
    }
    std::printf("[ipc] probe with %d processes: %s\n", n, bad ? "FAILED" : "done");
    return bad;
}
