// peer.hpp -- PeerComm: the exchange step as a PUSH over directly mapped peer memory (one process per GPU, or several
// contexts in one process), with device-side arrival flags instead of RCCL's ring or the in-process group's host barriers.
//
// Why: the exchange of the row-sharded W side moves 17 MB + 16 MB per outer iteration at the headline shape while a rank's
// compute is ~0.35 ms.  A ring collective is bound by ONE xGMI link (7 steps of 2 MB each way); xGMI is point-to-point, so a
// direct exchange -- every rank stores piece q straight into rank q's memory, all 7 links at once -- moves the same bytes in
// 1/7 of the time ("all-7-links direct reduce-scatter + all-gather ~ 0.03 ms").  It also removes the
// ~1300 latency-bound small all-reduces per outer iteration of ALSPGrad's line searches: their 3 doubles travel as tagged
// 16-byte granules inside the decision kernel itself (pgrad.hpp), no collective launch at all.
//
// Every rank owns a WINDOW (device memory exported with hipIpcGetMemHandle; peers in the same process use the pointer itself):
//   header   flag[q]   (u32) = sequence number of the last collective whose data from rank q has completely arrived HERE
//            abort     (u32) = a wait of SOME rank timed out.  The rank that times out stores 1 into the abort word of EVERY mapped
//                              window (system scope), so all ranks fail together: from then on every wait returns at once, pushes and
//                              signals are skipped (a rank that runs ahead must not overwrite the parity slots of peers that are
//                              still reading), and the host raises a communicator error at its next synchronisation point
//                              (health()).  The word is STICKY: a context whose exchange timed out keeps failing -- destroy it and
//                              build a new communicator (the peers' state is unknown after a time-out; nothing here could resync it).
//            tiny inbox      : { value, tag } granules of the in-kernel all-reduce (TinyAR)
//   data     2 parities x nranks slots of slot_bytes: collective s uses parity s & 1, slot q receives rank q's contribution
// A collective (or a group of them: PeerComm::group_start / group_end) with sequence number s is
//   push    : every rank stores its contribution(s) into slot[s & 1][rank] of the destination windows (write-through stores)
//   signal  : behind the push: flag[rank] = s in every destination window (system-scope store behind a release fence)
//   wait    : the SAME single-block launch then spins until flag[q] >= s for every q (bounded: wall clock), system-scope acquire
//   read    : the consumer sums / copies the slots in RANK ORDER q = 0 .. n-1 -- the order LocalComm's kernels use, so the two
//             transports agree bit for bit (tests/test_gpu_peer.py).
// Only single-block launches ever spin: a grid-wide consumer that polled would occupy the whole GPU while the peers it waits for
// cannot be scheduled when several ranks share one device (the 1-GPU test box; measured with scripts/kbench/ipc_probe.hip: 11 ms
// per exchange and time-outs with 8 processes, against 5-30 us for single-block spinners).
// Buffer re-use: rank A's push of collective s+2 into parity s & 1 of rank B happens after A passed the wait of s+1, i.e. after B
// pushed s+1, which B's stream issued behind its own read of s: two parities suffice as long as every rank runs every collective
// in the same order on one stream (they do: the solver enqueues the same sequence on every rank).
//
// Anything that does not fit a slot (and every collective in the pipelined mode's second stream) goes to the wrapped `base`
// transport (RCCL / the in-process group), which also carried the bootstrap: the decision depends on sizes only, so every rank
// takes the same route.  The reference has no distributed path: no reference counterpart.
#pragma once
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "comm.hpp"

namespace nmfx {

constexpr size_t PEER_HDR_BYTES = 16384;
constexpr size_t PEER_FLAG_OFF = 0;       // u32 flag[LOCAL_MAX_RANKS]
constexpr size_t PEER_ABORT_OFF = 256;    // u32
constexpr size_t PEER_TINYCNT_OFF = 512;  // u64: in-kernel all-reduces EXECUTED by this rank so far (the next one's tag - 1)
constexpr size_t PEER_TINY_OFF = 1024;    // TinyGran inbox[2][LOCAL_MAX_RANKS][PEER_TINY_MAX]
constexpr int PEER_TINY_MAX = 8;

struct __attribute__((aligned(16))) TinyGran {
    double v;
    unsigned long long tag;
};

typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
// system-scope (sc0 sc1) accesses through buffer descriptors: the compiler keeps track of their s_waitcnt
__device__ __forceinline__ __amdgpu_buffer_rsrc_t peer_rsrc(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, -1, 0x00020000);
}
constexpr int PEER_AUX = 17;   // gfx940+: bit 0 = sc0, bit 4 = sc1

__device__ __forceinline__ unsigned peer_load_u32(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void peer_store_u32(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// a wait timed out: every rank must fail, not only this one (a peer that was merely slow would otherwise find later flags and sum
// slot data of later collectives with its own abort word still 0)
__device__ __forceinline__ void peer_abort_all(const PeerWin &w, int n) {
    for (int q = 0; q < n; ++q)
        if (w.p[q]) peer_store_u32(reinterpret_cast<unsigned *>(w.p[q] + PEER_ABORT_OFF), 1u);
}
__device__ __forceinline__ bool peer_aborted(const unsigned char *mine) {
    return peer_load_u32(reinterpret_cast<const unsigned *>(mine + PEER_ABORT_OFF)) != 0;
}

// The in-kernel all-reduce (sum) of nval <= PEER_TINY_MAX doubles, called by ALL threads of ONE block (blockDim >= 64) with the
// block's values in vals[] (every thread holds the same values, or at least thread j holds vals[j]): each value goes to every
// rank's inbox as ONE 16-byte { value, tag } store -- value and tag land together, so the tag is the arrival flag -- and the
// ranks' values are added in rank order.  Returns the sums in vals[] to every thread.  ~3-5 us between processes on one GPU
// (ipc_probe), against a collective launch + a local reduction launch in front of it.
// The tag is a DEVICE-side count of executed calls (kept in the rank's own window): kernels that return early (a line search that is
// already over, an inner iteration behind a converged one) do not consume a tag, so the tags of the calls that do run are
// consecutive on every rank and two inbox parities suffice -- rank A can be at most one call ahead of rank B, because finishing
// call c needs B's granule of call c, which B stores only after it finished call c - 1.
// The granule is self-validating: its tag word carries the call count (low half) AND a hash of the value's bits (high half).  The
// 16-byte store / load is observed untorn on gfx950, but the memory model only promises single-copy atomicity up to 8 bytes: a reader
// that saw the new tag beside the value of call c - 2 would feed a wrong sum into one rank's line-search decision and the ranks would
// diverge silently -- with the hash such a granule simply does not match yet.
__device__ __forceinline__ unsigned long long tiny_tag(unsigned long long count, double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned h = ((unsigned)b ^ (unsigned)(b >> 32)) * 0x9E3779B1u + 0x7F4A7C15u;
    return (count & 0xffffffffull) | ((unsigned long long)h << 32);
}
__device__ __forceinline__ bool tiny_allreduce(const TinyAR &t, double *vals, int nval, double *sm /* >= PEER_TINY_MAX doubles of LDS */) {
    if (t.n <= 1) return true;
    const int tid = threadIdx.x;
    __shared__ int ok_sm;
    __shared__ unsigned long long tag_sm;
    if (tid == 0) {
        unsigned long long *cnt = reinterpret_cast<unsigned long long *>(t.win.p[t.rank] + PEER_TINYCNT_OFF);
        const unsigned long long c = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1;
        __hip_atomic_store(cnt, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        tag_sm = c;
        ok_sm = 1;
    }
    __syncthreads();
    const unsigned long long tag = tag_sm;
    const int par = (int)(tag & 1);
    for (int e = tid; e < t.n * nval; e += blockDim.x) {
        const int q = e / nval, j = e % nval;
        TinyGran g{vals[j], tiny_tag(tag, vals[j])};
        TinyGran *dst = reinterpret_cast<TinyGran *>(t.win.p[q] + PEER_TINY_OFF) + ((par * LOCAL_MAX_RANKS + t.rank) * PEER_TINY_MAX + j);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_t, g), peer_rsrc(dst), 0, 0, PEER_AUX);
    }
    __syncthreads();
    if (tid < nval) {
        double s = 0.0;
        const unsigned *abortw = reinterpret_cast<const unsigned *>(t.win.p[t.rank] + PEER_ABORT_OFF);
        for (int q = 0; q < t.n; ++q) {
            const TinyGran *src = reinterpret_cast<const TinyGran *>(t.win.p[t.rank] + PEER_TINY_OFF) + ((par * LOCAL_MAX_RANKS + q) * PEER_TINY_MAX + tid);
            const unsigned long long t0 = wall_clock64();
            int spins = 0;
            while (true) {
                const TinyGran g = __builtin_bit_cast(TinyGran, __builtin_amdgcn_raw_buffer_load_b128(peer_rsrc(src), 0, 0, PEER_AUX));
                if (g.tag == tiny_tag(tag, g.v)) { s += g.v; break; }   // (a torn granule -- new tag, old value -- fails the check and is polled again)
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 255) == 0 && (peer_load_u32(abortw) != 0 || wall_clock64() - t0 > t.timeout_ticks)) {
                    peer_abort_all(t.win, t.n);
                    ok_sm = 0;
                    break;
                }
            }
        }
        sm[tid] = s;
    }
    __syncthreads();
    for (int j = 0; j < nval; ++j) vals[j] = sm[j];
    const bool ok = ok_sm != 0;
    __syncthreads();
    return ok;
}

// ---- collective kernels ------------------------------------------------------------------------------------------------
// System-scope 16-byte accesses at base + off.  The descriptor is built from a WAVE-UNIFORM base (a kernel argument plus block / loop
// terms), the lane's part travels as the 32-bit offset operand.  (Round 6: a descriptor built from a per-lane pointer --
// peer_rsrc(d + i * 16), as these kernels had it -- is not uniform, and the compiler then wraps every access in a waterfall loop: one
// trip per DISTINCT lane value, i.e. 64 serial one-lane accesses per wave instruction.  The 16 MB push of the 8-rank shard shape took
// 42 us that way.)  Buffers beyond 2 GiB are walked in windows of 2 GiB with the window start folded into the base.
constexpr size_t PEER_WINDOW = (size_t)1 << 31;
__device__ __forceinline__ v4u_t peer_ld16(const unsigned char *base_uniform, uint32_t off) {
    return __builtin_amdgcn_raw_buffer_load_b128(peer_rsrc(base_uniform), (int)off, 0, PEER_AUX);
}
__device__ __forceinline__ void peer_st16(unsigned char *base_uniform, uint32_t off, v4u_t v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, peer_rsrc(base_uniform), (int)off, 0, PEER_AUX);
}
// push: bytes [q * src_stride, q * src_stride + bytes) of src  ->  window q at dst_off   (blockIdx.y = destination rank)
static __global__ __launch_bounds__(256) void peer_push_kernel(PeerWin w, size_t dst_off, const unsigned char *src, size_t bytes, size_t src_stride, int rank) {
    if (peer_aborted(w.p[rank])) return;   // after a time-out nothing is pushed any more (the peers may still read the slots)
    const int q = blockIdx.y;
    const unsigned char *s = src + (size_t)q * src_stride;
    unsigned char *d = w.p[q] + dst_off;
    if ((((uintptr_t)s | (uintptr_t)d | bytes) & 15) == 0) {
        const uint32_t stride = gridDim.x * blockDim.x;
        for (size_t w0 = 0; w0 < bytes; w0 += PEER_WINDOW) {
            const uint32_t nv = (uint32_t)(((bytes - w0 < PEER_WINDOW) ? bytes - w0 : PEER_WINDOW) / 16);
            const v4u_t *sw = reinterpret_cast<const v4u_t *>(s + w0);
            unsigned char *dw = d + w0;
            uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
            for (; i + 3 * stride < nv; i += 4 * stride) {   // four loads in flight per lane
                const v4u_t v0 = sw[i], v1 = sw[i + stride], v2 = sw[i + 2 * stride], v3 = sw[i + 3 * stride];
                peer_st16(dw, i * 16u, v0);
                peer_st16(dw, (i + stride) * 16u, v1);
                peer_st16(dw, (i + 2 * stride) * 16u, v2);
                peer_st16(dw, (i + 3 * stride) * 16u, v3);
            }
            for (; i < nv; i += stride) peer_st16(dw, i * 16u, sw[i]);
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < bytes; i += (size_t)gridDim.x * blockDim.x)
            __builtin_amdgcn_raw_buffer_store_b8(s[i], peer_rsrc(d + i), 0, 0, PEER_AUX);
    }
}
// wait (ONE block): until every rank's flag has reached seq; bounded by the wall clock (100 MHz ticks)
// signal != 0: the launch signals first (round 6: signal and wait of a collective in ONE launch -- a launch boundary less per exchange,
// ~3 us at the 8-rank shard shape; still a single block that spins, so ranks that share a device keep scheduling each other)
static __global__ void peer_wait_kernel(PeerWin w, int rank, int n, unsigned seq, unsigned long long timeout_ticks, int signal, int wait) {
    const int q = threadIdx.x;
    unsigned char *mine = w.p[rank];
    unsigned *abortw = reinterpret_cast<unsigned *>(mine + PEER_ABORT_OFF);
    if (signal && !peer_aborted(mine)) {   // (after a time-out nothing is signalled any more: a peer that has not failed yet times out on this rank's flag)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        if (q < n) peer_store_u32(reinterpret_cast<unsigned *>(w.p[q] + PEER_FLAG_OFF) + rank, seq);
    }
    if (!wait) return;
    if (q < n) {
        const unsigned *f = reinterpret_cast<const unsigned *>(mine + PEER_FLAG_OFF) + q;
        const unsigned long long t0 = wall_clock64();
        int spins = 0;
        while ((int)(peer_load_u32(f) - seq) < 0) {
            __builtin_amdgcn_s_sleep(4);
            if ((++spins & 63) == 0 && (peer_load_u32(abortw) != 0 || wall_clock64() - t0 > timeout_ticks)) {
                peer_abort_all(w, n);
                break;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}
// read, reducing: dst[i] = slot_0[i] (+|max) slot_1[i] ... in rank order; slot_q = mine + region_off + q * slot_bytes
template <typename T>
__global__ __launch_bounds__(256) void peer_reduce_kernel(T *dst, const unsigned char *mine, size_t region_off, size_t slot_bytes, int n, size_t count, int max_op) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned char *b = mine + region_off + i * sizeof(T);
        T s = __hip_atomic_load(reinterpret_cast<const T *>(b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int q = 1; q < n; ++q) {
            const T v = __hip_atomic_load(reinterpret_cast<const T *>(b + (size_t)q * slot_bytes), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            s = max_op ? (v > s ? v : s) : (s + v);
        }
        dst[i] = s;
    }
}
// read, gathering: dst[q * bytes + i] = slot_q[i]
static __global__ __launch_bounds__(256) void peer_gather_kernel(unsigned char *dst, const unsigned char *mine, size_t region_off, size_t slot_bytes, size_t bytes) {
    const int q = blockIdx.y;
    const unsigned char *s = mine + region_off + (size_t)q * slot_bytes;
    unsigned char *d = dst + (size_t)q * bytes;
    if ((((uintptr_t)s | (uintptr_t)d | bytes) & 15) == 0) {
        const uint32_t stride = gridDim.x * blockDim.x;
        for (size_t w0 = 0; w0 < bytes; w0 += PEER_WINDOW) {
            const uint32_t nv = (uint32_t)(((bytes - w0 < PEER_WINDOW) ? bytes - w0 : PEER_WINDOW) / 16);
            const unsigned char *sw = s + w0;
            v4u_t *dw = reinterpret_cast<v4u_t *>(d + w0);
            uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
            for (; i + 3 * stride < nv; i += 4 * stride) {
                const v4u_t v0 = peer_ld16(sw, i * 16u), v1 = peer_ld16(sw, (i + stride) * 16u), v2 = peer_ld16(sw, (i + 2 * stride) * 16u),
                            v3 = peer_ld16(sw, (i + 3 * stride) * 16u);
                dw[i] = v0; dw[i + stride] = v1; dw[i + 2 * stride] = v2; dw[i + 3 * stride] = v3;
            }
            for (; i < nv; i += stride) dw[i] = peer_ld16(sw, i * 16u);
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < bytes; i += (size_t)gridDim.x * blockDim.x)
            d[i] = __builtin_amdgcn_raw_buffer_load_b8(peer_rsrc(s + i), 0, 0, PEER_AUX);
    }
}

// ---- bootstrap handle (what the host ships between the ranks: nmfx_comm_p2p_export / nmfx_comm_p2p_attach) ---------------
struct PeerHandle {
    uint64_t magic;        // 'NMFXP2P1'
    int32_t pid, device;
    uint64_t ptr, bytes;
    uint64_t slot_bytes;
    hipIpcMemHandle_t ipc;
};
static_assert(sizeof(PeerHandle) <= 128, "NMFX_P2P_HANDLE_BYTES");
constexpr uint64_t PEER_MAGIC = 0x4e4d465850325031ull;

struct PeerComm : Comm {
    Comm *base;                       // bootstrap transport; carries whatever does not fit a slot (owned)
    int dev;
    unsigned char *mine = nullptr;    // this rank's window
    size_t win_bytes = 0, slot_bytes = 0;
    PeerWin win;                      // every rank's window as mapped HERE
    bool opened[LOCAL_MAX_RANKS] = {};
    bool attached = false, sim = false;
    unsigned seq = 0;
    unsigned long long timeout_ticks;
    // group state
    struct Op { int kind; void *buf; const void *send; size_t count; int ct; bool max_op; size_t off; };
    std::vector<Op> ops;
    bool grouping = false, group_fallback = false;
    size_t group_off = 0;
    long long n_peer = 0, n_base = 0;   // collectives served by the windows / handed to `base` (introspection: nmfx_comm_p2p_stats)

    PeerComm(Comm *base_, int device_, size_t slot_bytes_) : base(base_), dev(device_) {
        rank = base->rank;
        nranks = base->nranks;
        sim = std::strcmp(base->transport(), "sim") == 0;
        slot_bytes = (slot_bytes_ + 255) / 256 * 256;
        win_bytes = PEER_HDR_BYTES + 2 * (size_t)nranks * slot_bytes;
        double tmo_s = 30.0;
        if (const char *e = std::getenv("NMFX_P2P_TIMEOUT_S")) tmo_s = std::max(0.01, std::atof(e));
        timeout_ticks = (unsigned long long)(tmo_s * 1e8);   // wall_clock64: 100 MHz
        if (const char *e = dev_env("NMFX_P2P_TINY")) tiny_enabled = std::atoi(e) != 0;
        // uncached: remote stores and local reads both bypass the (non-coherent) L2s; NMFX_P2P_MEM=finegrained|plain for experiments
        const char *kind = dev_env("NMFX_P2P_MEM");
        hipError_t e;
        if (kind && std::strcmp(kind, "plain") == 0) e = hipMalloc(reinterpret_cast<void **>(&mine), win_bytes);
        else e = hipExtMallocWithFlags(reinterpret_cast<void **>(&mine), win_bytes, (kind && std::strcmp(kind, "finegrained") == 0) ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
        if (e != hipSuccess) throw CommError{std::string("peer window allocation failed: ") + hipGetErrorString(e)};
        if (hipMemset(mine, 0, win_bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) throw CommError{"peer window memset failed"};
        for (int q = 0; q < LOCAL_MAX_RANKS; ++q) win.p[q] = nullptr;
        win.p[rank] = mine;
    }
    ~PeerComm() override {
        for (int q = 0; q < nranks; ++q)
            if (opened[q] && win.p[q]) (void)hipIpcCloseMemHandle(win.p[q]);
        if (mine) (void)hipFree(mine);
        delete base;
    }
    const char *transport() const override { return "peer"; }

    void export_handle(void *out128) {
        PeerHandle h;
        std::memset(&h, 0, sizeof h);
        h.magic = PEER_MAGIC;
        h.pid = (int32_t)getpid();
        h.device = dev;
        h.ptr = (uint64_t)(uintptr_t)mine;
        h.bytes = win_bytes;
        h.slot_bytes = slot_bytes;
        const hipError_t e = hipIpcGetMemHandle(&h.ipc, mine);
        if (e != hipSuccess) { (void)hipGetLastError(); std::memset(&h.ipc, 0, sizeof h.ipc); }   // same-process peers need the pointer only
        std::memset(out128, 0, 128);
        std::memcpy(out128, &h, sizeof h);
    }
    void attach(const void *all_handles) {
        for (int q = 0; q < nranks; ++q) {
            PeerHandle h;
            std::memcpy(&h, reinterpret_cast<const unsigned char *>(all_handles) + (size_t)q * 128, sizeof h);
            if (h.magic != PEER_MAGIC) throw CommError{"peer attach: bad handle (rank " + std::to_string(q) + ")"};
            if (h.bytes != win_bytes || h.slot_bytes != slot_bytes) throw CommError{"peer attach: the ranks' windows differ in size (different p, k or dtype?)"};
            if (q == rank) continue;
            if (h.pid == (int32_t)getpid()) {
                if (h.device != dev) {
                    const hipError_t e = hipDeviceEnablePeerAccess(h.device, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) throw CommError{"hipDeviceEnablePeerAccess failed"};
                    (void)hipGetLastError();
                }
                win.p[q] = reinterpret_cast<unsigned char *>((uintptr_t)h.ptr);
            } else {
                void *ptr = nullptr;
                const hipError_t e = hipIpcOpenMemHandle(&ptr, h.ipc, hipIpcMemLazyEnablePeerAccess);
                if (e != hipSuccess) throw CommError{std::string("hipIpcOpenMemHandle failed: ") + hipGetErrorString(e)};
                win.p[q] = reinterpret_cast<unsigned char *>(ptr);
                opened[q] = true;
            }
        }
        attached = true;
    }
    // Back to the wrapped transport: unmap the peers' windows, every later collective goes to `base`.  For the case where mapping
    // failed on SOME rank (no IPC support, no peer access): the ranks where it succeeded must not keep routing through windows the
    // others do not use -- their collectives would no longer pair up.  The caller makes this collective (nmfx/dist.py::attach_p2p).
    void detach() {
        for (int q = 0; q < nranks; ++q) {
            if (opened[q] && win.p[q]) (void)hipIpcCloseMemHandle(win.p[q]);
            opened[q] = false;
            if (q != rank) win.p[q] = nullptr;
        }
        attached = false;
    }
    static void ck(hipError_t e, const char *what) {
        if (e != hipSuccess) throw CommError{std::string(what) + ": " + hipGetErrorString(e)};
    }
    // has a wait of this rank timed out?  (host-side, at the solver's synchronisation points)
    void health() override {
        unsigned a = 0;
        ck(hipMemcpy(&a, mine + PEER_ABORT_OFF, 4, hipMemcpyDeviceToHost), "hipMemcpy");
        if (a) throw CommError{"peer exchange timed out: a rank did not arrive within NMFX_P2P_TIMEOUT_S (default 30 s)"};
    }
    bool tiny_enabled = true;   // NMFX_P2P_TINY=0: the line-search scalars as ordinary (window) all-reduces of 3 doubles -- A/B of the in-kernel form
    bool tiny_capable() const override { return attached && tiny_enabled; }
    TinyAR tiny() override {
        TinyAR t;
        t.win = win;
        t.rank = rank;
        t.n = sim ? 1 : nranks;
        t.timeout_ticks = timeout_ticks;
        return t;
    }

    static unsigned grid_for(size_t items) {
        const size_t b = (items + 255) / 256;
        return (unsigned)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
    }
    size_t region(unsigned s) const { return PEER_HDR_BYTES + (size_t)(s & 1) * (size_t)nranks * slot_bytes; }

    bool fits(size_t bytes) const { return attached && group_off + ((bytes + 255) / 256 * 256) <= slot_bytes; }
    void enqueue(Op op, size_t bytes, hipStream_t s) {
        op.off = group_off;
        group_off += (bytes + 255) / 256 * 256;
        ops.push_back(op);
        if (!grouping) flush(s);
    }
    void group_start() override {
        grouping = true;
        group_fallback = false;
        group_off = 0;
        ops.clear();
        group_stream = nullptr;
    }
    void group_end() override {
        grouping = false;
        if (group_fallback) base->group_end();
        if (!ops.empty()) flush(group_stream);
        group_off = 0;
    }
    hipStream_t group_stream = nullptr;
    // members of a group that do not fit go to `base` inside base's own group
    void to_base_begin() {
        if (grouping && !group_fallback) { base->group_start(); group_fallback = true; }
        ++n_base;
    }

    // one sequence number for all queued operations: push all, signal, wait, read all
    void flush(hipStream_t s) {
        ck(hipSetDevice(dev), "hipSetDevice");
        const unsigned sq = ++seq;
        const size_t reg = region(sq);
        const int n = nranks;
        for (const Op &o : ops) {
            const size_t dst = reg + (size_t)rank * slot_bytes + o.off;
            const size_t es = ct_size(o.ct);
            if (o.kind == 0) {          // all-reduce: the whole buffer to every rank
                const size_t b = o.count * es;
                hipLaunchKernelGGL(peer_push_kernel, dim3(grid_for(b / 16 + 1), n), dim3(256), 0, s, win, dst, reinterpret_cast<const unsigned char *>(o.buf), b, (size_t)0, rank);
            } else if (o.kind == 1) {   // reduce-scatter: piece q to rank q
                const size_t b = o.count * es;
                hipLaunchKernelGGL(peer_push_kernel, dim3(grid_for(b / 16 + 1), n), dim3(256), 0, s, win, dst, reinterpret_cast<const unsigned char *>(o.send), b, b, rank);
            } else if (o.kind == 2) {   // all-gather: the chunk to every rank
                const size_t b = o.count * es;
                hipLaunchKernelGGL(peer_push_kernel, dim3(grid_for(b / 16 + 1), n), dim3(256), 0, s, win, dst, reinterpret_cast<const unsigned char *>(o.send), b, (size_t)0, rank);
            }                           // kind 3: pushed by the producer itself (direct_* below)
        }
        hipLaunchKernelGGL(peer_wait_kernel, dim3(1), dim3(64), 0, s, win, rank, n, sq, timeout_ticks, 1, sim ? 0 : 1);
        for (const Op &o : ops) {
            const size_t src = reg + o.off;
            if (o.kind == 0 || o.kind == 1) {
                void *dstp = o.buf;
                if (o.ct == CT_F32)
                    hipLaunchKernelGGL(peer_reduce_kernel<float>, dim3(grid_for(o.count)), dim3(256), 0, s, reinterpret_cast<float *>(dstp), mine, src, slot_bytes, n, o.count, o.max_op ? 1 : 0);
                else if (o.ct == CT_F64)
                    hipLaunchKernelGGL(peer_reduce_kernel<double>, dim3(grid_for(o.count)), dim3(256), 0, s, reinterpret_cast<double *>(dstp), mine, src, slot_bytes, n, o.count, o.max_op ? 1 : 0);
                else throw CommError{"byte reductions are not defined"};
            } else if (o.kind == 2) {
                const size_t b = o.count * ct_size(o.ct);
                hipLaunchKernelGGL(peer_gather_kernel, dim3(grid_for(b / 16 + 1), n), dim3(256), 0, s, reinterpret_cast<unsigned char *>(o.buf), mine, src, slot_bytes, b);
            }
        }
        ck(hipGetLastError(), "peer collective launch");
        n_peer += (long long)ops.size();
        ops.clear();
        group_off = 0;
    }

    // One sequence counter and two parities are only safe on ONE stream (header: "every rank runs every collective in the same order
    // on one stream").  The windows therefore serve the stream of their FIRST collective -- the solver's main stream (`home`;
    // Solver::attach sets it explicitly) -- and nothing else: a collective on any other stream (the pipelined mode's second stream)
    // goes to `base`, grouped or not.
    hipStream_t home = nullptr;
    bool home_set = false;
    bool on_home(hipStream_t s) {
        if (!home_set) { home = s; home_set = true; }
        return s == home;
    }
    void all_reduce(void *buf, size_t count, int ct, bool max_op, hipStream_t s) override {
        const size_t es = ct_size(ct);
        if (!attached || !on_home(s)) { to_base_begin(); base->all_reduce(buf, count, ct, max_op, s); return; }
        if (fits(count * es)) {
            group_stream = s;
            enqueue(Op{0, buf, nullptr, count, ct, max_op, 0}, count * es, s);
            return;
        }
        // larger than what is left of a slot (the packed all-reduce of the replicated-W mode): whatever the group holds so far
        // goes first, then the buffer travels in slot-sized pieces, each a collective of its own
        if (!ops.empty()) flush(group_stream ? group_stream : s);
        const size_t per = slot_bytes / 256 * 256 / es;
        for (size_t o = 0; o < count; o += per) {
            const size_t c = (count - o < per) ? (count - o) : per;
            ops.push_back(Op{0, reinterpret_cast<unsigned char *>(buf) + o * es, nullptr, c, ct, max_op, 0});
            flush(s);
        }
    }
    void reduce_scatter(const void *send, void *recv, size_t recvcount, int ct, hipStream_t s) override {
        if (!fits(recvcount * ct_size(ct)) || !on_home(s)) { to_base_begin(); base->reduce_scatter(send, recv, recvcount, ct, s); return; }
        group_stream = s;
        enqueue(Op{1, recv, send, recvcount, ct, false, 0}, recvcount * ct_size(ct), s);
    }
    void all_gather(const void *send, void *recv, size_t sendcount, int ct, hipStream_t s) override {
        if (!fits(sendcount * ct_size(ct)) || !on_home(s)) { to_base_begin(); base->all_gather(send, recv, sendcount, ct, s); return; }
        group_stream = s;
        enqueue(Op{2, recv, send, sendcount, ct, false, 0}, sendcount * ct_size(ct), s);
    }

    // broadcast: the root pushes slot-sized pieces into slot[root] of every window, everybody signals and waits (one sequence number
    // per piece, like every collective here), the others copy the piece out
    void broadcast(void *buf, size_t bytes, int root, hipStream_t s) override {
        if (!attached || !on_home(s)) { to_base_begin(); base->broadcast(buf, bytes, root, s); return; }
        if (!ops.empty()) flush(group_stream ? group_stream : s);
        ck(hipSetDevice(dev), "hipSetDevice");
        const size_t per = slot_bytes / 256 * 256;
        for (size_t o = 0; o < bytes; o += per) {
            const size_t b = (bytes - o < per) ? (bytes - o) : per;
            const unsigned sq = ++seq;
            const size_t reg = region(sq);
            unsigned char *piece = reinterpret_cast<unsigned char *>(buf) + o;
            if (rank == root)
                hipLaunchKernelGGL(peer_push_kernel, dim3(grid_for(b / 16 + 1), nranks), dim3(256), 0, s, win, reg + (size_t)root * slot_bytes, piece, b, (size_t)0, rank);
            hipLaunchKernelGGL(peer_wait_kernel, dim3(1), dim3(64), 0, s, win, rank, nranks, sq, timeout_ticks, 1, sim ? 0 : 1);
            if (rank != root)
                hipLaunchKernelGGL(peer_gather_kernel, dim3(grid_for(b / 16 + 1), 1), dim3(256), 0, s, piece, mine, reg + (size_t)root * slot_bytes, slot_bytes, b);
            ck(hipGetLastError(), "peer broadcast launch");
            ++n_peer;
        }
    }

    // ---- producer-side push (the GEMM-fused exchange) --------------------------------------------------------------------
    // Inside a group: reserve `bytes` of this group's slot space for data a producer kernel stores ITSELF -- its epilogue writes the
    // piece for rank q to  direct_dst(q, off)  (peer memory, write-through stores) -- and the consumer, launched behind group_end()
    // (which signals and waits; no push / read launch for this member), reads the n contributions from  direct_src(q, off).
    // Both pointers belong to the sequence number group_end() is about to take: ask for them BEFORE group_end().
    size_t direct_reserve(size_t bytes) {
        if (!grouping || !fits(bytes)) return (size_t)-1;
        const size_t off = group_off;
        group_off += (bytes + 255) / 256 * 256;
        ops.push_back(Op{3, nullptr, nullptr, 0, CT_BYTE, false, off});
        return off;
    }
    size_t direct_off(size_t off) const { return region(seq + 1) + (size_t)rank * slot_bytes + off; }   // offset of direct_dst inside every window
    unsigned char *direct_dst(int q, size_t off) const { return win.p[q] + region(seq + 1) + (size_t)rank * slot_bytes + off; }
    const unsigned char *direct_src(int q, size_t off) const { return mine + region(seq + 1) + (size_t)q * slot_bytes + off; }
    // PULL form of a reserved member: every rank stores its contribution into ITS OWN window (direct_dst(rank, off)), the consumers behind
    // group_end() read rank q's contribution out of q's window: pull_src(q, off).  Buffer re-use is the push form's argument with the
    // roles swapped: a rank overwrites its slot of parity s & 1 for collective s + 2 after it passed the wait of s + 1, i.e. after
    // every peer signalled s + 1, which a peer's stream issues behind its own reads of s.
    // (timing stand-in: every window is this rank's, slot q holds nothing -- the own slot stands in for all of them)
    const unsigned char *pull_src(int q, size_t off) const {
        const int qq = sim ? rank : q;
        return win.p[qq] + region(seq + 1) + (size_t)qq * slot_bytes + off;
    }
};

}  // namespace nmfx
