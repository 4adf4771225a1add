// comm.hpp -- the exchange step of the sharded path behind one small interface, with two transports:
//
//   RcclComm   one process per GPU (the production / bench.py layout): RCCL collectives over xGMI on the solver's streams.
//   LocalComm  several contexts inside ONE process (one host thread per context; the contexts may sit on different GPUs --
//              peer-to-peer loads over xGMI -- or on the SAME GPU): every collective is a small hand-written kernel that
//              reads the peers' buffers directly, ordered by hipEvents across the ranks' streams.  No device-side flags or
//              spinning: the ordering is host-enqueued (record -> host barrier -> hipStreamWaitEvent), so a rank that fails
//              cannot hang a GPU, only a host barrier, which times out.  On the 1-GPU test box this runs the sharded code of
//              libnmfx itself with 2, 4 or 8 ranks on one device (tests/test_gpu_localcomm.py); reductions add the ranks'
//              contributions in rank order, so results are deterministic and identical on every rank.
//
// The reference has no distributed path: this file has no reference counterpart.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

namespace nmfx {

// Development switches (A/B measurements, tests of alternative kernels; the NMFX_* names read through this function) are honoured
// only when NMFX_DEV=1 is set as well: a caller's environment cannot change what the library computes by accident.  The one
// run-time setting a deployment may want, NMFX_P2P_TIMEOUT_S, is read directly.
static inline const char *dev_env(const char *name) {
    static const bool on = [] { const char *e = std::getenv("NMFX_DEV"); return e != nullptr && e[0] == '1'; }();
    return on ? std::getenv(name) : nullptr;
}

struct CommError {
    std::string msg;
};

// element type tag of a collective
enum : int { CT_F32 = 0, CT_F64 = 1, CT_BYTE = 2 };
static inline size_t ct_size(int ct) { return ct == CT_F32 ? 4 : (ct == CT_F64 ? 8 : 1); }

constexpr int LOCAL_MAX_RANKS = 16;

// every rank's exchange window as mapped into THIS rank's address space (peer.hpp)
struct PeerWin {
    unsigned char *p[LOCAL_MAX_RANKS];
};
// what a one-block kernel needs to all-reduce a handful of doubles INSIDE itself (peer.hpp: tiny_allreduce); n <= 1: nothing to do
struct TinyAR {
    PeerWin win;
    int rank = 0, n = 1;
    unsigned long long timeout_ticks = 0;   // wall_clock64 ticks (100 MHz)
};

struct Comm {
    int rank = 0, nranks = 1;
    virtual ~Comm() {}
    virtual const char *transport() const = 0;
    // did an exchange of this rank time out on the device?  (throws CommError; transports without device-side waits: no-op)
    virtual void health() {}
    // in-kernel all-reduce of a few doubles (peer transport)
    virtual bool tiny_capable() const { return false; }
    virtual TinyAR tiny() { return TinyAR(); }
    // in-place element-wise sum / max over the ranks
    virtual void all_reduce(void *buf, size_t count, int ct, bool max_op, hipStream_t s) = 0;
    // recv[i] = sum_q send_q[rank*recvcount + i]
    virtual void reduce_scatter(const void *send, void *recv, size_t recvcount, int ct, hipStream_t s) = 0;
    // recv[q*sendcount + i] = send_q[i]
    virtual void all_gather(const void *send, void *recv, size_t sendcount, int ct, hipStream_t s) = 0;
    // buf of rank `root` -> buf of every rank, bit for bit (the winner of solve_replicates! travelling to every GPU)
    virtual void broadcast(void *buf, size_t bytes, int root, hipStream_t s) = 0;
    virtual void group_start() {}
    virtual void group_end() {}
};

// ---------------------------------------------------------------------------------------------------------------- RCCL
struct RcclFailure {
    ncclResult_t e;
    const char *what;
};
#define NMFX_RCCL(x)                                          \
    do {                                                      \
        ncclResult_t _e = (x);                                \
        if (_e != ncclSuccess) throw RcclFailure{_e, #x};     \
    } while (0)

struct RcclComm : Comm {
    ncclComm_t comm = nullptr;
    RcclComm(const void *uid_bytes, int rank_, int nranks_) {
        ncclUniqueId id;
        std::memcpy(&id, uid_bytes, sizeof id);
        NMFX_RCCL(ncclCommInitRank(&comm, nranks_, id, rank_));
        rank = rank_;
        nranks = nranks_;
    }
    ~RcclComm() override {
        if (comm) (void)ncclCommDestroy(comm);
    }
    const char *transport() const override { return "rccl"; }
    static ncclDataType_t dt(int ct) { return ct == CT_F32 ? ncclFloat : (ct == CT_F64 ? ncclDouble : ncclInt8); }
    void all_reduce(void *buf, size_t count, int ct, bool max_op, hipStream_t s) override {
        NMFX_RCCL(ncclAllReduce(buf, buf, count, dt(ct), max_op ? ncclMax : ncclSum, comm, s));
    }
    void reduce_scatter(const void *send, void *recv, size_t recvcount, int ct, hipStream_t s) override {
        NMFX_RCCL(ncclReduceScatter(send, recv, recvcount, dt(ct), ncclSum, comm, s));
    }
    void all_gather(const void *send, void *recv, size_t sendcount, int ct, hipStream_t s) override {
        NMFX_RCCL(ncclAllGather(send, recv, sendcount, dt(ct), comm, s));
    }
    void broadcast(void *buf, size_t bytes, int root, hipStream_t s) override {
        NMFX_RCCL(ncclBroadcast(buf, buf, bytes, ncclInt8, root, comm, s));
    }
    void group_start() override { NMFX_RCCL(ncclGroupStart()); }
    void group_end() override { NMFX_RCCL(ncclGroupEnd()); }
};

// ------------------------------------------------------------------------------------------------ timing stand-in
// "Rank r of n" with NO peers: every collective moves the bytes it would receive device-locally (reduce-scatter: own chunk;
// all-gather: own chunk into every slot; all-reduce: nothing).  Numerically meaningless -- it exists so that the per-rank
// COMPUTE of the sharded path at an n-rank shard shape can be timed on a 1-GPU box (bench.py --sim-ranks).
// dst[q * nvec + i] = src[i] for every q != skip: the all-gather stand-in as ONE launch (a collective is one launch; the 7 separate
// hipMemcpyAsync calls this used to be cost 19 us of launch latency at the 8-rank shard shape for 14 MB of copies)
static __global__ __launch_bounds__(256) void sim_replicate_kernel(uint4 *dst, const uint4 *src, size_t nvec, int n, int skip) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nvec * (size_t)n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t q = e / nvec, i = e % nvec;
        if ((int)q != skip) dst[e] = src[i];
    }
}

struct SimComm : Comm {
    SimComm(int rank_, int nranks_) { rank = rank_; nranks = nranks_; }
    const char *transport() const override { return "sim"; }
    void all_reduce(void *, size_t, int, bool, hipStream_t) override {}
    void reduce_scatter(const void *send, void *recv, size_t recvcount, int ct, hipStream_t s) override {
        const size_t b = recvcount * ct_size(ct);
        (void)hipMemcpyAsync(recv, reinterpret_cast<const char *>(send) + (size_t)rank * b, b, hipMemcpyDeviceToDevice, s);
    }
    void all_gather(const void *send, void *recv, size_t sendcount, int ct, hipStream_t s) override {
        const size_t b = sendcount * ct_size(ct);
        const char *own = reinterpret_cast<const char *>(send);
        const char *r0 = reinterpret_cast<const char *>(recv);
        if (b % 16 == 0 && ((uintptr_t)send % 16) == 0 && ((uintptr_t)recv % 16) == 0) {
            const int skip = (own >= r0 && own < r0 + b * (size_t)nranks && (size_t)(own - r0) % b == 0) ? (int)((size_t)(own - r0) / b) : -1;   // in-place: the own chunk is already there
            const size_t nvec = b / 16, items = nvec * (size_t)nranks;
            const unsigned grid = (unsigned)std::max<size_t>(1, std::min<size_t>((items + 255) / 256, 2048));
            hipLaunchKernelGGL(sim_replicate_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<uint4 *>(recv), reinterpret_cast<const uint4 *>(send), nvec, nranks, skip);
            return;
        }
        for (int q = 0; q < nranks; ++q) {
            char *dst = reinterpret_cast<char *>(recv) + (size_t)q * b;
            if (dst != send) (void)hipMemcpyAsync(dst, send, b, hipMemcpyDeviceToDevice, s);   // in-place all-gather: the own chunk is already there
        }
    }
    void broadcast(void *, size_t, int, hipStream_t) override {}
};

// "rank r of n" with NO transport of its own: the base of a PeerComm that must serve every collective from its windows
// (nmfx_comm_init_p2p: several processes on ONE device -- where RCCL refuses duplicate GPUs -- or a deployment without RCCL)
struct NoComm : Comm {
    NoComm(int rank_, int nranks_) { rank = rank_; nranks = nranks_; }
    const char *transport() const override { return "none"; }
    [[noreturn]] static void fail() { throw CommError{"this collective cannot be served by the peer windows (other stream or not attached) and the communicator has no base transport"}; }
    void all_reduce(void *, size_t, int, bool, hipStream_t) override { fail(); }
    void reduce_scatter(const void *, void *, size_t, int, hipStream_t) override { fail(); }
    void all_gather(const void *, void *, size_t, int, hipStream_t) override { fail(); }
    void broadcast(void *, size_t, int, hipStream_t) override { fail(); }
};

// --------------------------------------------------------------------------------------------------- in-process group

struct PeerPtrs {
    const void *p[LOCAL_MAX_RANKS];
};

// dst[i] = sum_q src_q[off + i] (q ascending), i < count
template <typename T> __global__ void local_reduce_kernel(T *dst, PeerPtrs src, int n, size_t off, size_t count, int max_op) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        T s = reinterpret_cast<const T *>(src.p[0])[off + i];
        for (int q = 1; q < n; ++q) {
            const T v = reinterpret_cast<const T *>(src.p[q])[off + i];
            s = max_op ? (v > s ? v : s) : (s + v);
        }
        dst[i] = s;
    }
}
// dst[q*count + i] = src_q[i]   (16-byte words when everything is aligned, bytes otherwise)
template <typename W> __global__ void local_gather_kernel(W *dst, PeerPtrs src, int n, size_t count) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < count * n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t q = e / count, i = e % count;
        dst[e] = reinterpret_cast<const W *>(src.p[q])[i];
    }
}

struct LocalGroup {
    int n;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool broken = false;
    const void *slot[LOCAL_MAX_RANKS] = {};
    hipEvent_t ready[LOCAL_MAX_RANKS] = {}, done[LOCAL_MAX_RANKS] = {};
    int device[LOCAL_MAX_RANKS] = {};
    std::atomic<int> attached{0};
    bool orphaned = false;   // nmfx_local_group_destroy was called while contexts were still attached: the last one to detach frees the group
    static std::mutex &lifetime_mu() { static std::mutex m; return m; }
    // owner's release (nmfx_local_group_destroy): immediate when no context is attached, deferred to the last detach otherwise
    static void release(LocalGroup *g) {
        bool now;
        {
            std::lock_guard<std::mutex> lk(lifetime_mu());
            g->orphaned = true;
            now = g->attached.load() == 0;
        }
        if (now) delete g;
    }
    static void detach(LocalGroup *g) {
        bool now;
        {
            std::lock_guard<std::mutex> lk(lifetime_mu());
            now = (g->attached.fetch_sub(1) == 1) && g->orphaned;
        }
        if (now) delete g;
    }
    explicit LocalGroup(int n_) : n(n_) {}
    ~LocalGroup() {
        for (int i = 0; i < LOCAL_MAX_RANKS; ++i) {
            if (ready[i]) (void)hipEventDestroy(ready[i]);
            if (done[i]) (void)hipEventDestroy(done[i]);
        }
    }
    // host barrier over the n rank threads; a missing rank (it failed) breaks the group for everyone after `timeout_s`
    void barrier(double timeout_s = 120.0) {
        std::unique_lock<std::mutex> lk(mu);
        if (broken) throw CommError{"local group is broken (a rank failed earlier)"};
        const uint64_t gen = generation;
        if (++arrived == n) {
            arrived = 0;
            ++generation;
            cv.notify_all();
            return;
        }
        const bool ok = cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return generation != gen || broken; });
        if (!ok || broken) {
            broken = true;
            cv.notify_all();
            throw CommError{"local group barrier timed out: not every rank reached the collective"};
        }
    }
};

struct LocalComm : Comm {
    LocalGroup *g;
    int dev;
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    LocalComm(LocalGroup *g_, int rank_, int device_) : g(g_), dev(device_) {
        rank = rank_;
        nranks = g->n;
        // a rank that attaches again (a context re-using the slot) replaces the slot's events instead of leaking them
        if (g->ready[rank]) { (void)hipEventDestroy(g->ready[rank]); g->ready[rank] = nullptr; }
        if (g->done[rank]) { (void)hipEventDestroy(g->done[rank]); g->done[rank] = nullptr; }
        if (hipEventCreateWithFlags(&g->ready[rank], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g->done[rank], hipEventDisableTiming) != hipSuccess)
            throw CommError{"hipEventCreate failed"};
        g->device[rank] = device_;
        g->barrier();   // every rank has published its device
        for (int q = 0; q < nranks; ++q)
            if (g->device[q] != dev) {
                const hipError_t e = hipDeviceEnablePeerAccess(g->device[q], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) throw CommError{"hipDeviceEnablePeerAccess failed"};
                (void)hipGetLastError();
            }
        g->attached.fetch_add(1);   // last: a constructor that throws has no destructor to detach again
    }
    ~LocalComm() override {
        if (scratch) (void)hipFree(scratch);
        LocalGroup::detach(g);
    }
    const char *transport() const override { return "local"; }

    void need_scratch(size_t bytes) {
        if (bytes <= scratch_bytes) return;
        if (scratch) (void)hipFree(scratch);
        if (hipMalloc(&scratch, bytes) != hipSuccess) throw CommError{"hipMalloc (collective scratch) failed"};
        scratch_bytes = bytes;
    }
    static void ck(hipError_t e, const char *what) {
        if (e != hipSuccess) throw CommError{std::string(what) + ": " + hipGetErrorString(e)};
    }
    // publish `mine`, make every rank's stream wait until all ranks' data is ready; returns the peers' pointers
    PeerPtrs exchange_begin(const void *mine, hipStream_t s) {
        g->slot[rank] = mine;
        ck(hipEventRecord(g->ready[rank], s), "hipEventRecord");
        g->barrier();
        PeerPtrs pp;
        for (int q = 0; q < LOCAL_MAX_RANKS; ++q) pp.p[q] = (q < nranks) ? g->slot[q] : nullptr;
        for (int q = 0; q < nranks; ++q)
            if (q != rank) ck(hipStreamWaitEvent(s, g->ready[q], 0), "hipStreamWaitEvent");
        return pp;
    }
    // after the reading kernel: nobody may overwrite a buffer a peer is still reading
    void exchange_end(hipStream_t s) {
        ck(hipEventRecord(g->done[rank], s), "hipEventRecord");
        g->barrier();
        for (int q = 0; q < nranks; ++q)
            if (q != rank) ck(hipStreamWaitEvent(s, g->done[q], 0), "hipStreamWaitEvent");
    }
    static unsigned grid_for(size_t count) {
        const size_t b = (count + 255) / 256;
        return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
    }
    template <typename T> void reduce_typed(T *dst, const PeerPtrs &pp, size_t off, size_t count, bool max_op, hipStream_t s) {
        hipLaunchKernelGGL(local_reduce_kernel<T>, dim3(grid_for(count)), dim3(256), 0, s, dst, pp, nranks, off, count, max_op ? 1 : 0);
        ck(hipGetLastError(), "local_reduce_kernel");
    }
    void reduce_any(void *dst, const PeerPtrs &pp, size_t off, size_t count, int ct, bool max_op, hipStream_t s) {
        if (ct == CT_F32) reduce_typed(reinterpret_cast<float *>(dst), pp, off, count, max_op, s);
        else if (ct == CT_F64) reduce_typed(reinterpret_cast<double *>(dst), pp, off, count, max_op, s);
        else throw CommError{"byte reductions are not defined"};
    }
    void all_reduce(void *buf, size_t count, int ct, bool max_op, hipStream_t s) override {
        ck(hipSetDevice(dev), "hipSetDevice");
        need_scratch(count * ct_size(ct));
        const PeerPtrs pp = exchange_begin(buf, s);
        reduce_any(scratch, pp, 0, count, ct, max_op, s);          // full sum into private scratch ...
        exchange_end(s);                                           // ... every rank is done reading ...
        ck(hipMemcpyAsync(buf, scratch, count * ct_size(ct), hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");   // ... publish
    }
    void reduce_scatter(const void *send, void *recv, size_t recvcount, int ct, hipStream_t s) override {
        ck(hipSetDevice(dev), "hipSetDevice");
        const PeerPtrs pp = exchange_begin(send, s);
        reduce_any(recv, pp, (size_t)rank * recvcount, recvcount, ct, false, s);
        exchange_end(s);
    }
    void all_gather(const void *send, void *recv, size_t sendcount, int ct, hipStream_t s) override {
        ck(hipSetDevice(dev), "hipSetDevice");
        const PeerPtrs pp = exchange_begin(send, s);
        const size_t bytes = sendcount * ct_size(ct);
        bool al16 = (bytes % 16 == 0) && ((uintptr_t)recv % 16 == 0);
        for (int q = 0; q < nranks; ++q) al16 = al16 && ((uintptr_t)pp.p[q] % 16 == 0);
        if (al16) {
            typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
            hipLaunchKernelGGL(local_gather_kernel<v4u_t>, dim3(grid_for(bytes / 16 * nranks)), dim3(256), 0, s,
                               reinterpret_cast<v4u_t *>(recv), pp, nranks, bytes / 16);
        } else {
            hipLaunchKernelGGL(local_gather_kernel<unsigned char>, dim3(grid_for(bytes * nranks)), dim3(256), 0, s,
                               reinterpret_cast<unsigned char *>(recv), pp, nranks, bytes);
        }
        ck(hipGetLastError(), "local_gather_kernel");
        exchange_end(s);
    }
    void broadcast(void *buf, size_t bytes, int root, hipStream_t s) override {
        ck(hipSetDevice(dev), "hipSetDevice");
        const PeerPtrs pp = exchange_begin(buf, s);
        if (rank != root) ck(hipMemcpyAsync(buf, pp.p[root], bytes, hipMemcpyDefault, s), "hipMemcpyAsync (broadcast)");
        exchange_end(s);
    }
};

}  // namespace nmfx
