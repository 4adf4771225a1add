// solver.hpp -- device-resident NMF solver behind the C ABI (include/nmfx.h).
//
// One Solver<T> = one GPU's share of one problem (column shard of X and H, replica of W).
// It replaces the reference's per-solve state objects and the nmf_skeleton! loop
// (src/common.jl:45-89): the loop runs as a stream of kernel launches with a
// device-side stop flag; the host polls that flag every `check_every` iterations.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <algorithm>
#include <chrono>
#include <map>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/nmfx.h"
#include "comm.hpp"
#include "peer.hpp"
#include "gemm_mfma.hpp"
#include "gemm_bf16x3.hpp"
#include "gemm_stream.hpp"
#include "kernels.hpp"
#include "cd.hpp"
#include "chol.hpp"

namespace nmfx {
struct PgState;

struct HipError {
    hipError_t e;
    const char *what;
    int line;
};
#define HIP_TRY(x)                                                  \
    do {                                                            \
        hipError_t _e = (x);                                        \
        if (_e != hipSuccess) throw HipError{_e, #x, __LINE__};     \
    } while (0)

struct RcclError {
    ncclResult_t e;
    const char *what;
    int line;
};
#define RCCL_TRY(x)                                                 \
    do {                                                            \
        ncclResult_t _e = (x);                                      \
        if (_e != ncclSuccess) throw RcclError{_e, #x, __LINE__};   \
    } while (0)

struct StatusError {
    int status;
    std::string msg;
};

static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

struct SolverBase {
    std::string err;
    int dtype = 0;
    virtual ~SolverBase() {}
    virtual void set_X(const void *X, int64_t ldx, bool on_device) = 0;
    virtual void set_factors(const void *W, const void *H) = 0;
    virtual void get_factors(void *W, void *H) = 0;
    virtual void iterate(int alg, const nmfx_opts &o, nmfx_result *out, double *trace) = 0;
    virtual void subsolve(int which, const nmfx_opts &o, nmfx_result *out) = 0;
    virtual void comm_init(const void *uid, int rank, int nranks) = 0;
    virtual void comm_init_local(LocalGroup *group, int rank) = 0;
    virtual void comm_set_mode(int mode) = 0;
    virtual void spa_init(int warm_sweeps, int64_t *anchors_out, int64_t *unsolved_out) = 0;
    virtual void pdsolve_host(int right, const void *A_host, const void *B_host, double lambda, void *X_host, bool clamp) = 0;
    virtual void comm_init_sim(int rank, int nranks) = 0;
    virtual void comm_init_p2p(int rank, int nranks) = 0;
    virtual void p2p_export(void *handle_out) = 0;
    virtual void p2p_attach(const void *all_handles) = 0;
    virtual void p2p_stats(long long *served_by_windows, long long *served_by_base) = 0;
    virtual double objective(int alg, const nmfx_opts &o) = 0;
    virtual bool check_nonneg(int which) = 0;
    virtual void randinit(uint64_t seed, bool normalize, bool zeroh, int64_t h_col_offset) = 0;
    virtual void solve_replicates(int alg, const nmfx_opts &o, int replicates, uint64_t seed, bool zeroh, int64_t h_col_offset,
                                  void *W_host, void *H_host, nmfx_result *out, int *best) = 0;
    virtual int get_iter_trace(double *elapsed, double *relchange, int count) = 0;
    virtual void rsvd_begin(uint64_t seed, int64_t h_col_offset, int power_iters, void *C_host) = 0;
    virtual void rsvd_finish(const void *Ub_host, const void *s_host, void *U_out, void *Vt_out) = 0;
    virtual void nndsvd_init(const void *U_host, const void *s_host, const void *V_host, int variant, bool zeroh, uint64_t seed,
                             int64_t n_total) = 0;
    virtual void profile_enable(int mode) = 0;
    virtual void set_final_objective(bool on) = 0;
    virtual int profile_get(nmfx_kernel_stat *out, int max_entries) = 0;
};

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t count = 0;
    void alloc(size_t n) {
        release();
        count = n;
        if (n == 0) return;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p), n * sizeof(T)));
        // hipMemset is stream-ordered on the NULL stream, which the solver's non-blocking streams do not wait for: without the
        // synchronisation a buffer that is allocated lazily (Q, the work arrays, the exchange buffers) could be zeroed AFTER
        // the first kernel wrote to it (seen once, with four contexts sharing one GPU: a wiped Q sent the KL objective to inf)
        HIP_TRY(hipMemset(p, 0, n * sizeof(T)));
        HIP_TRY(hipStreamSynchronize(nullptr));
    }
    void ensure(size_t n) {
        if (n > count) alloc(n);
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        count = 0;
    }
    ~DevBuf() { release(); }
};

template <typename T> class Solver : public SolverBase {
  public:
    using M = Mfma<T>;
    static constexpr int BK = M::BK;

    Solver(int64_t p_, int64_t n_, int64_t k_, int device_) : p(p_), n(n_), k(k_), device(device_) {
        dtype = sizeof(T) == 4 ? NMFX_F32 : NMFX_F64;
        HIP_TRY(hipSetDevice(device));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        num_cu = prop.multiProcessorCount;
        HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        // ProjectedALS runs its single-workgroup factorisations UNDER the big products (projals_impl.hpp): the products launch
        // `chol_slots` blocks short of two per CU, a high-priority side stream owns the half-empty CUs that leaves.
        // NMFX_CHOL_SLOTS=0 turns it off (factorisations between the products, as in round 1).
        if (const char *e = dev_env("NMFX_CHOL_SLOTS")) chol_slots = std::max(0, std::min(128, std::atoi(e)));
        if (const char *e = dev_env("NMFX_FORCE_SHARDED")) force_sharded = std::atoi(e) != 0;
        if (const char *e = dev_env("NMFX_CD_LDS")) cd_force_lds = std::atoi(e) != 0;
        if (const char *e = dev_env("NMFX_CD_BLOCKED")) cd_blocked = std::atoi(e) != 0 ? 1 : 0;
        if (const char *e = dev_env("NMFX_RS_FUSED")) rs_fused_enabled = std::atoi(e) != 0;
        if (const char *e = dev_env("NMFX_STREAM_WH")) stream_wh = std::atoi(e) != 0;
        if (const char *e = dev_env("NMFX_SMALLK")) smallk_enabled = std::atoi(e) != 0;
        if (const char *e = dev_env("NMFX_DIV_FUSED")) div_fused = std::atoi(e) != 0;
        if (const char *e = dev_env("NMFX_STATS_FUSED")) stats_fused_enabled = std::atoi(e) != 0;
        if (const char *e = std::getenv("NMFX_DIV_IEEE")) div_ieee = std::atoi(e) != 0;
        if (const char *e = dev_env("NMFX_K_GRANULE")) k_granule = (std::atoi(e) == 128) ? 128 : 64;
        if (const char *e = dev_env("NMFX_POTRS")) { potrs_enabled = std::atoi(e) != 0; potrs_iter = std::atoi(e) == 1; }
        if (const char *e = dev_env("NMFX_POTRS_STRIP")) strip_enabled = std::atoi(e) != 0;
        if (const char *e = dev_env("NMFX_POTRF_REG")) potrf_reg_enabled = std::atoi(e) != 0;
        if (const char *e = dev_env("NMFX_CHOL_UNROLLED")) chol_unrolled = std::atoi(e) != 0;
        if (const char *e = dev_env("NMFX_FUSE_GRAM")) fuse_gram = std::atoi(e) != 0;
        if (const char *e = dev_env("NMFX_UNSPLIT")) unsplit_enabled = std::atoi(e) != 0;
        if (const char *e = dev_env("NMFX_DIRECT_MAX_KTILES")) direct_max_ktiles = std::max(32, std::atoi(e));
        if (const char *e = dev_env("NMFX_DIRECT_LONG")) direct_long_local = std::atoi(e) != 0;
        if (const char *e = dev_env("NMFX_STOP_SUMS_V1")) stop_sums_v1 = std::atoi(e) != 0;
        if (const char *e = dev_env("NMFX_PROJALS_XT")) xht_images = std::atoi(e) != 0;
        if (const char *e = dev_env("NMFX_CHOL_UNDER_US")) chol_under_min_us = std::atof(e);
        if (const char *e = std::getenv("NMFX_XT")) xt_enabled = std::atoi(e) != 0;
        if (const char *e = dev_env("NMFX_W_BLOCKED")) blk_enabled = std::atoi(e) != 0;
        if (const char *e = dev_env("NMFX_P2P_PULL")) peer_pull_enabled = std::atoi(e) != 0;
        if (const char *e = dev_env("NMFX_DEFER_CHECK")) defer_enabled = std::atoi(e) != 0;
        HIP_TRY(hipEventCreate(&ev_beg));
        HIP_TRY(hipEventCreate(&ev_end));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ctrl), sizeof(Ctrl)));
        HIP_TRY(hipMemset(ctrl, 0, sizeof(Ctrl)));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ctrl_host), sizeof(Ctrl)));
        layout(256);
    }

    // Padded extents and every buffer whose size depends on them.  row_mult = the multiple P is rounded up to: 256 for one
    // GPU; lcm(256, 128 * nranks) once a communicator is attached, so that the row-sharded W side
    // gets whole 128-row tiles per rank.
    void layout(int64_t row_mult) {
        P = round_up(p, row_mult);
        N = round_up(n, 256);
        // K: multiples of 64 (round 4; multiples of 128 above 64 before: k = 129 ... 192 paid for 256 components in every product).
        // K % 128 == 0 keeps every fused path (Gram riding in the big launches, the persistent W*H kernel, the factorisations under the
        // products, the fused row-sharded step); the 64-granular sizes in between (192, 320, ...) run the general sequence on
        // 128 x 64 / 64 x 128 / 64 x 64 tiles.  NMFX_K_GRANULE=128 restores the old padding (A/B).
        K = (k <= 64) ? 64 : round_up(k, k_granule);
        // the fused epilogues address a wave tile with 32-bit byte offsets (gemm_mfma.hpp, "Epilogue addressing"):
        // 256 rows x leading dimension must stay below 4 GiB.  Leading dimensions are P (W, X, Q) and K (H, Gram).
        if ((uint64_t)std::max(P, K) * sizeof(T) * 256 >= (1ull << 32))
            throw StatusError{NMFX_ERR_UNSUPPORTED, "p (or k) too large: 256 * leading dimension * sizeof(T) must be < 2^32"};
        X.alloc((size_t)P * N);
        for (int i = 0; i < 2; ++i) { W[i].alloc((size_t)P * K); H[i].alloc((size_t)K * N); }
        // H-side numerator and Gram are ONE K x (N+K) matrix: [ W'X | W'W ] is produced by a single GEMM launch
        hside.alloc((size_t)K * (N + K));
        numH_p = hside.p;
        gramW_p = hside.p + (size_t)K * N;
        // W-side numerator, Gram and multdiv's rowsum(H) live in ONE buffer: it is the packed
        // all-reduce payload of the column-sharded path  [ X_g H_g' | H_g H_g' | rowsum(H_g) ]
        pack.alloc((size_t)P * K + (size_t)K * K + (size_t)K);
        numW_p = pack.p;
        gramH_p = pack.p + (size_t)P * K;
        sH_p = gramH_p + (size_t)K * K;
        // split-K slabs: sized for the largest (splits x output) product any GEMM of the path can ask for
        s_h = pick_splits((int)(N / 128) * (int)((K + 127) / 128), P);
        s_w = pick_splits((int)(P / 128) * (int)((K + 127) / 128), N);
        s_gw = pick_splits((int)((K + 127) / 128) * (int)((K + 127) / 128), P);
        s_gh = pick_splits((int)((K + 127) / 128) * (int)((K + 127) / 128), N);
        if (const char *e = dev_env("NMFX_BIG_SPLITS")) {   // (experiment: force the split count of the two big products)
            const int f = std::atoi(e);
            if (f >= 1 && (P / BK) % f == 0 && (N / BK) % f == 0) { s_h = f; s_w = f; }
        }
        // slab buffer = [ big-GEMM slabs | Gram slabs ]: the two live side by side so the update GEMM can consume
        // the un-reduced numerator slabs directly in its epilogue
        // (x PIPE_C: the pipelined exchange launches the big products per row super-chunk, each with its own split-K slabs)
        const size_t h_region = (size_t)s_h * K * N * PIPE_C;
        const size_t w_region = (size_t)std::max(s_w, pick_splits((int)(P / PIPE_C / 128) * (int)((K + 127) / 128), N)) * P * K;
        slab_w_off = h_region;
        gram_slab_off = h_region + w_region;
        // Gram slabs: split-K slabs of the stand-alone Gram launch, or tail pieces of the fused launch
        // (pieces <= blocks / tail tiles, see tail_piece)
        const size_t tail_tiles_total = std::max<size_t>(1, ((K + 127) / 128) * ((K + 127) / 128));
        const size_t max_pieces = (size_t)(2 * num_cu * 4) / tail_tiles_total + 2;
        max_gram_slabs = (int)std::max<size_t>((size_t)std::max(s_gw, s_gh), max_pieces) * PIPE_C;
        slabs.alloc(gram_slab_off + (size_t)max_gram_slabs * K * K);
        stat_chunks_w = (int)std::max<int64_t>(1, std::min<int64_t>(64, P / 1024));
        stat_chunks_h = (int)std::max<int64_t>(1, std::min<int64_t>(1024, N / 16));   // 4 chips' worth of workgroups for the column-chunked passes over H
        stat_part.alloc((size_t)std::max<int64_t>(std::max(stat_chunks_w, stat_chunks_h), std::max(N, P) / 16) * 3 * K);   // smallk: one partial per 16-wide stripe; multdiv: 3 values per (chunk, component)
        wstat.alloc((size_t)2 * K);
        hstat.alloc((size_t)4 * K);   // (x 2: by iteration parity while the stop rule is deferred, see defer_pending)
        svec.alloc((size_t)K);
        obj_part.alloc((size_t)2 * (P / 128) * (N / 128) + 4096);
        obj_extra.alloc(4);
        obj_final.alloc(1);
        for (auto &w : work) w.release();
        Q.release(); Wbest.release(); Hbest.release();
        Xt.release(); Ht[0].release(); Ht[1].release(); xt_valid = false;
        have_X = have_F = false;
        HIP_TRY(hipDeviceSynchronize());
    }

    ~Solver() override {
        (void)hipSetDevice(device);
        (void)hipStreamSynchronize(stream);
        delete comm;
        comm = nullptr;
        if (cstream) {
            (void)hipStreamSynchronize(cstream);
            for (int c = 0; c < PIPE_C; ++c) { (void)hipEventDestroy(ev_red[c]); (void)hipEventDestroy(ev_rs[c]); (void)hipEventDestroy(ev_pack[c]); (void)hipEventDestroy(ev_ag[c]); }
            (void)hipEventDestroy(ev_tail);
            (void)hipStreamDestroy(cstream);
        }
        if (fstream) {
            (void)hipStreamSynchronize(fstream);
            (void)hipEventDestroy(ev_fork);
            (void)hipEventDestroy(ev_join);
            (void)hipStreamDestroy(fstream);
        }
        for (auto &e : ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        if (pg_state) (void)hipFree(pg_state);
        if (pg_host) (void)hipHostFree(pg_host);
        if (ctrl) (void)hipFree(ctrl);
        if (ctrl_host) (void)hipHostFree(ctrl_host);
        for (hipEvent_t e : iter_events) (void)hipEventDestroy(e);
        (void)hipEventDestroy(ev_beg);
        (void)hipEventDestroy(ev_end);
        (void)hipStreamDestroy(stream);
    }

    // ---------------------------------------------------------------- data movement
    void set_X(const void *Xsrc, int64_t ldx, bool on_device) override {
        if (ldx < p) throw StatusError{NMFX_ERR_DIM_MISMATCH, "ldx < p"};
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipMemsetAsync(X.p, 0, X.count * sizeof(T), stream));
        HIP_TRY(hipMemcpy2DAsync(X.p, P * sizeof(T), Xsrc, ldx * sizeof(T), p * sizeof(T), n,
                                 on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        have_X = true;
        xt_valid = false;
    }

    void set_factors(const void *Wsrc, const void *Hsrc) override {
        HIP_TRY(hipSetDevice(device));
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(hipMemsetAsync(W[i].p, 0, W[i].count * sizeof(T), stream));
            HIP_TRY(hipMemsetAsync(H[i].p, 0, H[i].count * sizeof(T), stream));
        }
        HIP_TRY(hipMemcpy2DAsync(W[0].p, P * sizeof(T), Wsrc, p * sizeof(T), p * sizeof(T), k, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpy2DAsync(H[0].p, K * sizeof(T), Hsrc, k * sizeof(T), k * sizeof(T), n, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        wcur = hcur = 0;
        have_F = true;
        if (rsvd_ready == 1) rsvd_ready = 0;
    }

    void get_factors(void *Wdst, void *Hdst) override {
        HIP_TRY(hipSetDevice(device));
        if (Wdst) HIP_TRY(hipMemcpy2DAsync(Wdst, p * sizeof(T), W[wcur].p, P * sizeof(T), p * sizeof(T), k, hipMemcpyDeviceToHost, stream));
        if (Hdst) HIP_TRY(hipMemcpy2DAsync(Hdst, k * sizeof(T), H[hcur].p, K * sizeof(T), k * sizeof(T), n, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }

    void comm_init(const void *uid, int rank_, int nranks_) override {
        HIP_TRY(hipSetDevice(device));
        static_assert(sizeof(ncclUniqueId) <= NMFX_UNIQUE_ID_BYTES, "unique id size");
        attach(new RcclComm(uid, rank_, nranks_));
    }
    void comm_init_local(LocalGroup *group, int rank_) override {
        HIP_TRY(hipSetDevice(device));
        attach(new LocalComm(group, rank_, device));
    }
    void comm_init_sim(int rank_, int nranks_) override {
        HIP_TRY(hipSetDevice(device));
        attach(new SimComm(rank_, nranks_));
    }
    // Peer-to-peer exchange (peer.hpp).  comm_init_p2p: a communicator whose ONLY transport is the ranks' windows (the host ships
    // the 128-byte handles, as it ships RCCL's unique id); p2p_export on a context that already has a communicator (RCCL, the
    // in-process group, the timing stand-in) wraps it: the windows serve what fits, the wrapped transport the rest.
    void comm_init_p2p(int rank_, int nranks_) override {
        HIP_TRY(hipSetDevice(device));
        if (nranks_ < 1 || nranks_ > LOCAL_MAX_RANKS || rank_ < 0 || rank_ >= nranks_) throw StatusError{NMFX_ERR_BAD_ARG, "peer communicator: need 0 <= rank < nranks <= 16"};
        attach(new NoComm(rank_, nranks_));
    }
    // a slot holds one rank's contribution to the largest group of the W side's exchange: the Pc x K piece of the numerator (or of the
    // new W: the all-gather chunk), two k x k Grams, the k-vectors and the statistics
    size_t p2p_slot_bytes() const {
        const size_t piece = std::max((size_t)std::max<int64_t>(Pc, 0) * K * sizeof(T) + 4096, ag_chunk_bytes + 4096);
        // (+ the statistics tail of a blocked-residency chunk: up to 64 chunks of 2K partials)
        return piece + 2 * ((size_t)K * K * sizeof(T) + 256) + 8 * ((size_t)K * sizeof(double) + 256) + 4096 + (size_t)64 * 2 * K * sizeof(double);
    }
    void p2p_export(void *handle_out) override {
        HIP_TRY(hipSetDevice(device));
        if (!comm) throw StatusError{NMFX_ERR_STATE, "nmfx_comm_p2p_export needs a communicator (nmfx_comm_init / _init_local / _init_sim / _init_p2p first)"};
        if (nranks > LOCAL_MAX_RANKS) throw StatusError{NMFX_ERR_UNSUPPORTED, "peer exchange: at most 16 ranks"};
        PeerComm *pc = dynamic_cast<PeerComm *>(comm);
        if (!pc) {
            Comm *b = comm;
            pc = new PeerComm(b, device, p2p_slot_bytes());   // (throws before taking ownership of b only if the window cannot be allocated)
            pc->home = stream; pc->home_set = true;           // the windows serve the main stream's collectives only (peer.hpp)
            comm = pc;
        }
        pc->export_handle(handle_out);
    }
    void p2p_attach(const void *all_handles) override {
        HIP_TRY(hipSetDevice(device));
        PeerComm *pc = dynamic_cast<PeerComm *>(comm);
        if (!pc) throw StatusError{NMFX_ERR_STATE, "nmfx_comm_p2p_attach: call nmfx_comm_p2p_export first"};
        if (!all_handles) { HIP_TRY(hipStreamSynchronize(stream)); pc->detach(); return; }
        pc->attach(all_handles);
    }
    void p2p_stats(long long *w, long long *b) override {
        PeerComm *pc = dynamic_cast<PeerComm *>(comm);
        if (w) *w = pc ? pc->n_peer : 0;
        if (b) *b = pc ? pc->n_base : 0;
    }
    PeerComm *peer() const { PeerComm *pc = dynamic_cast<PeerComm *>(comm); return (pc && pc->attached) ? pc : nullptr; }
    // 0: row-sharded W side (reduce-scatter / all-gather, the default whenever the shapes allow it); 1: the replicated W
    // update behind one packed all-reduce (round-1 formulation, kept for comparison and as the fallback); 2: row-sharded with
    // the exchange pipelined against the big products (MultUpdate-MSE; the other algorithms run as in mode 0)
    void comm_set_mode(int mode) override {
        comm_mode = mode;
        // replicas: every solve is a one-GPU solve and must give the one-GPU bits -- undo the row padding attach() chose for a
        // row-sharded W side (split-K factors follow P)
        if (mode == NMFX_COMM_REPLICAS && P != round_up(p, 256)) {
            if (have_X) throw StatusError{NMFX_ERR_STATE, "nmfx_comm_set_mode(NMFX_COMM_REPLICAS) must precede nmfx_set_X when p is not a multiple of lcm(256, 128*nranks)"};
            HIP_TRY(hipSetDevice(device));
            layout(256);
        }
    }
    void attach(Comm *c) {
        delete comm;
        comm = c;
        rank = c->rank;
        nranks = c->nranks;
        if (sharded()) {
            // whole 128-row tiles per rank for the row-sharded W side
            int64_t m = 128 * (int64_t)nranks * PIPE_C, a = 256, b = m;   // x PIPE_C: whole tiles per rank and row super-chunk
            while (b) { const int64_t t = a % b; a = b; b = t; }
            const int64_t row_mult = 256 / a * m;
            if (round_up(p, row_mult) != P) {
                if (have_X) throw StatusError{NMFX_ERR_STATE, "nmfx_comm_init must be called before nmfx_set_X when p is not a multiple of lcm(256, 128*nranks)"};
                layout(row_mult);
            }
            Pc = P / nranks;
            row0 = (int64_t)rank * Pc;
            Rc = P / PIPE_C;
            Pcc = Pc / PIPE_C;
            ag_chunk_bytes = (size_t)Pc * K * sizeof(T) + (size_t)2 * K * sizeof(double);
            agc_bytes = (size_t)Pcc * K * sizeof(T) + (size_t)2 * K * sizeof(double);
            rs_out.alloc((size_t)Pc * K);
            ag_send.alloc(std::max(ag_chunk_bytes, agc_bytes * PIPE_C));
            ag_recv.alloc(std::max(ag_chunk_bytes, agc_bytes * PIPE_C) * (size_t)nranks);
            if (!cstream) {
                HIP_TRY(hipStreamCreateWithFlags(&cstream, hipStreamNonBlocking));
                for (int c = 0; c < PIPE_C; ++c) {
                    HIP_TRY(hipEventCreateWithFlags(&ev_red[c], hipEventDisableTiming));
                    HIP_TRY(hipEventCreateWithFlags(&ev_rs[c], hipEventDisableTiming));
                    HIP_TRY(hipEventCreateWithFlags(&ev_pack[c], hipEventDisableTiming));
                    HIP_TRY(hipEventCreateWithFlags(&ev_ag[c], hipEventDisableTiming));
                }
                HIP_TRY(hipEventCreateWithFlags(&ev_tail, hipEventDisableTiming));
            }
        }
    }

    // mode 0: off; 1: hipEvent pair around EVERY launch (each pair costs ~10 us of stream time: use for
    // per-kernel breakdowns, not for throughput); 2: only the dominant GEMM launches (the p*n*k products) and the
    // collectives, every 8th one -- the live roofline measurement of bench.py, < 0.5 % overhead on the timed region;
    // 3: the same launches, every 4th one (short runs: a 20-step line rests on 5 samples per kernel; every 2nd cost 0.075 ms per iteration);
    // 4: every 16th (ProjectedALS: a bracket on the main stream delays the factorisation stream's ordering events, ~0.4 ms per bracketed iteration).
    // nmfx_set_final_objective: off = nmfx_iterate leaves Result.objvalue NaN instead of evaluating it after the loop (a caller that
    // times K iterations -- bench.py -- asks nmfx_objective for it afterwards; the evaluation is one more p*n*k product)
    bool final_objective = true;
    void set_final_objective(bool on) override { final_objective = on; }
    void profile_enable(int mode) override {
        profiling = mode;
        records.clear();
        ev_used = 0;
        prof_seen.clear();
    }

    int profile_get(nmfx_kernel_stat *out, int max_entries) override {
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipStreamSynchronize(stream));
        std::map<std::string, nmfx_kernel_stat> agg;
        std::vector<std::string> order;
        for (auto &r : records) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, ev_pool[r.ev].first, ev_pool[r.ev].second));
            auto it = agg.find(r.name);
            if (it == agg.end()) {
                nmfx_kernel_stat s;
                std::memset(&s, 0, sizeof s);
                std::snprintf(s.name, sizeof s.name, "%s", r.name);
                it = agg.emplace(r.name, s).first;
                order.push_back(r.name);
            }
            it->second.ms_total += ms;
            it->second.launches += 1;
            it->second.flops += r.flops;
            it->second.bytes += r.bytes;
        }
        int cnt = 0;
        for (auto &nm : order) {
            if (cnt >= max_entries) break;
            out[cnt++] = agg[nm];
        }
        return cnt;
    }

    // ---------------------------------------------------------------- the loop
    void iterate(int alg, const nmfx_opts &o, nmfx_result *out, double *trace) override;
    void subsolve(int which, const nmfx_opts &o, nmfx_result *out) override;
    // nnmf front end on the device (frontend_impl.hpp)
    bool check_nonneg(int which) override;
    void randinit(uint64_t seed, bool normalize, bool zeroh, int64_t h_col_offset) override;
    void solve_replicates(int alg, const nmfx_opts &o, int replicates, uint64_t seed, bool zeroh, int64_t h_col_offset, void *W_host,
                          void *H_host, nmfx_result *out, int *best) override;
    void nndsvd_init(const void *U_host, const void *s_host, const void *V_host, int variant, bool zeroh, uint64_t seed,
                     int64_t n_total) override;
    void nndsvd_core(const T *Ud, int64_t ucs, int64_t uss, const T *Vd, int64_t vcs, int64_t vss, const T *sd, T *coef, int variant,
                     bool zeroh, uint64_t seed, int64_t n_total);
    void pdsolve_host(int right, const void *A_host, const void *B_host, double lambda, void *X_host, bool clamp) override;
    bool rsvd_cholqr2(T *Qbuf, T *tmp);   // rsvd_impl.hpp
    void spa_init(int warm_sweeps, int64_t *anchors_out, int64_t *unsolved_out) override;   // spa_impl.hpp
    DevBuf<long long> flag_ll;   // spa: the anchor indices
    DevBuf<int> spa_status;      // spa: per-column outcome of the active-set solve
    DevBuf<double> spa_tri;      // spa: packed triangles when they do not fit the LDS
    // randomized SVD of the resident X (rsvd_impl.hpp)
    void rsvd_begin(uint64_t seed, int64_t h_col_offset, int power_iters, void *C_host) override;
    void rsvd_finish(const void *Ub_host, const void *s_host, void *U_out, void *Vt_out) override;
    int rsvd_ready = 0;   // 1: Q, B resident (after begin); 2: U, s, V' resident (after finish)
    double objective(int alg, const nmfx_opts &o) override {
        require_ready();
        HIP_TRY(hipSetDevice(device));
        precision = o.precision;   // the opts of THIS call decide, not those of the last solve
        if (rsvd_ready == 1) rsvd_ready = 0;   // numH_p (the rsvd's B) is not touched, but keep the contract simple: begin/finish back to back
        enqueue_objective(alg, o, obj_final.p, nullptr);
        double v = 0.0;
        HIP_TRY(hipMemcpyAsync(&v, obj_final.p, sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return v;
    }

  private:
    int64_t p, n, k, P, N, K;
    int64_t k_granule = 64;
    int device, num_cu = 256;
    hipStream_t stream = nullptr;
    hipEvent_t ev_beg = nullptr, ev_end = nullptr;
    size_t potrf_lds_bytes() const {
        const size_t kp0 = (size_t)(k + 31) / 32 * 32, kps = kp0 > 64 ? kp0 - 32 : 32;
        return (size_t)(32 * 32 + 32 * kps) * sizeof(T);
    }
    DevBuf<T> X, Q, W[2], H[2], hside, slabs, svec, pack;
    // Second image of X, transposed (N x P, ld N), and of the current H (N x K, ld N): with them the X*H' product of the W side
    // (src/multupd.jl:109) contracts over the CONTIGUOUS index of both operands, like W'X does, and runs on the same kernel
    // instantiation (contraction-contiguous staging, k-loop unrolled by two) instead of the row-contiguous one, whose register
    // transposes cost ~50 % more issued instructions (profiles/r04_bench_multmse_c3.md: 118.9 M vs 79.0 M) and 4-5 % of time.
    // X' is built once per uploaded X (one transpose pass, + p*n elements of HBM: 1 GiB of 288 at the headline shape); H' is written by
    // the H update's own epilogue (EpiMultUpdate<T, 2>).  Float32, MultUpdate-MSE (general path); NMFX_XT=0 keeps the old product.
    DevBuf<T> Xt, Ht[2];
    bool xt_enabled = true, xt_valid = false, ht_active = false;
    bool want_xt() const { return xt_enabled && sizeof(T) == 4 && !use_bf16x3(); }
    void ensure_xt() {
        if (xt_valid || !want_xt()) return;
        if (Xt.count < (size_t)P * N) {
            // The image doubles the memory held for X (+ p*n elements).  It is an optimisation, so it must never be what makes a problem
            // that fits without it run out of memory later: it is only taken if, after it, there is still room for what the iteration
            // allocates lazily behind it (H' twice, the split-K slabs / work arrays: a few (p + n) k, bounded here by 8 (P + N) K elements
            // + 256 MiB); otherwise -- or if the allocation itself fails -- the row-contiguous product stays.
            size_t free_b = 0, total_b = 0;
            const size_t need = (size_t)P * N * sizeof(T), headroom = (size_t)8 * (size_t)(P + N) * K * sizeof(T) + ((size_t)256 << 20);
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < need + headroom) { (void)hipGetLastError(); xt_enabled = false; return; }
            T *q = nullptr;
            if (hipMalloc(reinterpret_cast<void **>(&q), need) != hipSuccess) { (void)hipGetLastError(); xt_enabled = false; return; }   // no room: keep the row-contiguous product
            Xt.release(); Xt.p = q; Xt.count = (size_t)P * N;
        }
        hipLaunchKernelGGL(transpose_kernel<T>, dim3((unsigned)((P / 64) * (N / 64))), dim3(256), 0, stream, Xt.p, N, X.p, P, P, N, (const int *)nullptr);
        HIP_TRY(hipGetLastError());
        xt_valid = true;
    }
    // H' of the current H (start of a solve; afterwards the update epilogue keeps it current)
    // (false: no room for H' -- the caller keeps the row-contiguous product)
    bool refresh_ht() {
        try {
            for (auto &h : Ht) h.ensure((size_t)N * K);
        } catch (const HipError &) {      // (DevBuf::alloc: hipMalloc failed)
            (void)hipGetLastError();
            for (auto &h : Ht) h.release();
            xt_enabled = false;
            return false;
        }
        hipLaunchKernelGGL(transpose_kernel<T>, dim3((unsigned)((K / 64) * (N / 64))), dim3(256), 0, stream, Ht[hcur].p, N, H[hcur].p, K, K, N, (const int *)nullptr);
        HIP_TRY(hipGetLastError());
        return true;
    }
    // H' of `Hp` for ONE X*H' product on the transposed images (the coordinate-descent updaters: their sweeps write H, not H'), or
    // nullptr: not Float32, sharded, or no room for the images.  One 2 K N element transpose pass (~10 us at 16384 columns, k = 256)
    // buys the contraction-contiguous kernel for the product: 1043 -> ~930 us at 16384 x 16384 (round 6).
    const T *ht_for(const T *Hp, const int *done, bool sharded_too = false) {
        if (!want_xt() || (sharded() && !sharded_too)) return nullptr;
        ensure_xt();
        if (!xt_valid) return nullptr;
        try {
            Ht[0].ensure((size_t)N * K);
        } catch (const HipError &) {
            (void)hipGetLastError();
            return nullptr;
        }
        hipLaunchKernelGGL(transpose_kernel<T>, dim3((unsigned)((K / 64) * (N / 64))), dim3(256), 0, stream, Ht[0].p, N, Hp, K, K, N, done);
        HIP_TRY(hipGetLastError());
        return Ht[0].p;
    }
    T *numH_p = nullptr, *gramW_p = nullptr;
    T *numW_p = nullptr, *gramH_p = nullptr, *sH_p = nullptr;
    DevBuf<T> potrf_panel;   // potrf's row panel when it does not fit the LDS (k > 1248 f32 / 608 f64)
    DevBuf<T> work[8];   // algorithm-specific scratch (projals factor/inverse, alspgrad G/Zn/Zp/D/GD)
    DevBuf<double> stat_part, wstat, hstat, obj_part, obj_extra, obj_final, trace_dev;
    Ctrl *ctrl = nullptr, *ctrl_host = nullptr;
    DevBuf<T> Wbest, Hbest;   // solve_replicates: the best replicate's factors
    DevBuf<int> flagbuf;
    DevBuf<double> nd_scratch;   // nndsvd_init: column norms and sum(X) partials
    DevBuf<double> dev_trace;    // per-iteration devmax of stop_condition when tracking
    std::vector<hipEvent_t> iter_events;
    std::vector<double> iter_elapsed, iter_relchange;
    int iter_trace_len = 0;
    int wcur = 0, hcur = 0;
    int s_h = 1, s_w = 1, s_gw = 1, s_gh = 1;
    int stat_chunks_w = 1, stat_chunks_h = 1;
    size_t gram_slab_off = 0;
    int max_gram_slabs = 1;
    int last_tiles_r = 1;   // r-tiles of the most recent GEMM launch (= chunks of its statistics partials)
    int last_blocks = 1;    // blocks of the most recent GEMM launch (= number of its objective partials)
    bool have_X = false, have_F = false;
    // Fusing the k x k Gram into the big GEMM launch adds (K/128)^2 tiles to a grid that otherwise fills the 512
    // block slots exactly (256 tiles x 2 splits @C3); as plain extra tiles the 8 Gram blocks form a second wave
    // (+50 %: 1505 vs 1010 us), so they are dealt out as a short second segment of EVERY block (see the kernel).
    bool fuse_gram = true;    // K % 128 == 0: Gram rides in the big GEMM launch as a balanced tail segment
    size_t slab_w_off = 0;    // W-side slab region
    Comm *comm = nullptr;
    int rank = 0, nranks = 1;
    // A communicator of ONE rank normally short-cuts to the single-GPU code (no collective is issued).  NMFX_FORCE_SHARDED=1
    // (read at construction) keeps the sharded code path -- reduce-scatter / all-gather / grouped all-reduces, the pipelined
    // exchange -- also for nranks == 1, so that every collective of the multi-GPU step executes on a 1-GPU box, under RCCL, as
    // an identity (tests/test_gpu_comm.py).
    bool force_sharded = false;
    // (NMFX_COMM_REPLICAS: the communicator only carries solve_replicates' fan-out; every solve is a plain one-GPU solve)
    bool sharded() const { return comm != nullptr && comm_mode != NMFX_COMM_REPLICAS && (nranks > 1 || force_sharded); }
    bool replicas_mode() const { return comm != nullptr && comm_mode == NMFX_COMM_REPLICAS && nranks > 1; }
    DevBuf<unsigned char> rep_buf;   // solve_replicates over the ranks: the replicates' result records on their way through the all-gather
    int comm_mode = 0;
    // row-sharded W side: this rank updates rows [row0, row0 + Pc) of W
    int64_t Pc = 0, row0 = 0;
    size_t ag_chunk_bytes = 0;
    DevBuf<T> rs_out;                       // reduce-scatter output: this rank's rows of the summed numerator (Pc x K, ld Pc)
    DevBuf<unsigned char> ag_send, ag_recv; // all-gather chunks: [ Pc x K piece of the new W | 2K doubles of column statistics ]
    // MultUpdate-MSE, row-sharded: the launches between the big products and the collectives fused (solver_impl.hpp)
    bool rs_fused_enabled = true;         // NMFX_RS_FUSED=0: the unfused sequence (pack / unpack / statistics as separate launches)
    bool force_quarter_tiles = false;     // set around a gemm() call: 64 x 64 tiles whatever the shape
    bool w_direct = false;                // times_ht stored the product straight into the blocked send buffer (no slabs)
    bool gramw_sharded_valid = false;     // gramW_p holds the all-reduced W'W of the current W (row-sharded fused step)
    bool w_defer_combine = false;         // times_ht: leave the split-K slabs and the Gram pieces to the caller's combine launch
    bool h_reduce_pair = false;           // wt_times: numerator and Gram combined by ONE launch
    int w_pieces = 1;                     // Gram tail pieces of the last fused X*H' launch
    int h_stat_chunks = 1;                // r-tiles of the last H update (chunks of its statistics partials)
    bool check_fused = false;             // the stop check of this iteration already ran inside stats_check_kernel
    // stats_sum_check's arrival ticket (kernels.hpp): the partial sums of the row-sharded step's stop statistics are taken by
    // STAT_BLOCKS blocks, the last one to arrive runs the rule
    static constexpr unsigned STAT_BLOCKS = 8;
    DevBuf<unsigned> stat_ticket_buf;
    unsigned *stat_ticket() { stat_ticket_buf.ensure(32); return stat_ticket_buf.p; }
    // The stop rule of the row-sharded fused step (blocked residency) DEFERRED into the next iteration's combine launch (kernels.hpp:
    // DeferChk): iteration t leaves its statistics in the tails of the blocked W buffer and sets defer_pending; the next combine launch
    // -- or, at the host's poll points and at the end of the solve, a launch of its own (defer_flush) -- runs the rule for t.
    // hstat is double-buffered by iteration parity while this is on (the combine launch that checks t writes t + 1's H statistics).
    // NMFX_DEFER_CHECK=0: the rule in a launch of its own behind every iteration (round 5; A/B).
    bool defer_enabled = true, defer_pending = false;
    long long defer_t = 0;
    nmfx_opts defer_opts;
    double *hstat_of(long long t) { return hstat.p + ((defer_enabled && (t & 1)) ? (size_t)2 * K : (size_t)0); }
    DeferChk<T> defer_args(unsigned nb) {
        DeferChk<T> dc;
        std::memset(&dc, 0, sizeof dc);
        if (nb == 0 || !defer_pending) return dc;
        dc.tails = reinterpret_cast<const double *>(Wblk[wb].p + (size_t)Pc * K * sizeof(T));
        dc.nchunks = nranks * blk_cpp; dc.grp = blk_cpp;
        dc.grp_stride = (int64_t)(blk_chunk / sizeof(double));
        dc.wstat = wstat.p; dc.ctrl = ctrl;
        dc.hstat = defer_opts.update_H ? hstat_of(defer_t) : (const double *)nullptr;
        dc.ticket = stat_ticket();
        dc.K = (int)K; dc.k = (int)k;
        dc.tol = (T)defer_opts.tol; dc.t = defer_t; dc.nb = nb;
        return dc;
    }
    void defer_flush() {
        if (!defer_pending) return;
        const DeferChk<T> dc = defer_args(STAT_BLOCKS);
        hipLaunchKernelGGL(stats_check_kernel<T>, dim3(STAT_BLOCKS), dim3(256), 0, stream, dc.tails, dc.nchunks, (int)K, wstat.p, ctrl, dc.hstat, (int)k, dc.tol, dc.t, 1,
                           done_flag(), dc.grp, dc.grp_stride, dc.ticket);
        HIP_TRY(hipGetLastError());
        defer_pending = false;
    }
    bool rs_fused() const { return rs_fused_enabled && row_sharded() && fuse_gram && K % 128 == 0 && !use_bf16x3(); }
    bool row_sharded() const { return sharded() && comm_mode != NMFX_COMM_REPLICATED_W && Pc > 0 && Pc % 128 == 0; }
    // Pipelined exchange (pipeline_impl.hpp; NMFX_COMM_PIPELINED, MultUpdate-MSE): the W side runs per row super-chunk, chunk
    // c's reduce-scatter on a second stream under chunk c+1's X*H' launch, its all-gather under the next iteration's W'X part.
    static constexpr int PIPE_C = 2;
    int64_t Rc = 0, Pcc = 0;              // rows per super-chunk, rows per (super-chunk, rank)
    size_t agc_bytes = 0;
    hipStream_t cstream = nullptr;        // the collectives of the pipelined mode
    // ProjectedALS hides its factorisations under a product only if the product is estimated at least this long (projals_impl.hpp).  Measured
    // (scripts/r06_step11.sh, k = 256, Float32, ms per iteration in stream order / under the products, with round 6's trtri): 4096^2
    // (57 us) 0.425 / 0.409, 8192 x 4096 (115) 0.541 / 0.583, 8192^2 (229) 0.800 / 0.686, 12288 x 8192 (344) 1.156 / 1.009; before the
    // trtri change 8192^2 was 0.861 / 1.695 and 8192 x 16384 (458) 1.332 / 1.145, 16384 x 12288 (687) 2.034 / 1.827
    double chol_under_min_us = 200.0;
    int chol_slots = 8;                   // block slots (half CUs) the big products of ProjectedALS leave to the factorisation stream
    bool short_grid = false;              // set around the products that must leave those slots
    int potrf_nt = 1024;                  // threads of the Cholesky workgroup (512 when it has to fit beside a GEMM block)
    hipStream_t fstream = nullptr;        // the factorisation stream (created by the first ProjectedALS iteration)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    void ensure_fstream() {
        if (fstream) return;
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIP_TRY(hipStreamCreateWithPriority(&fstream, hipStreamNonBlocking, hi));
        HIP_TRY(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    }
    // the k x k Gram products by their own launches (the overlapped ProjectedALS path needs them BEFORE the big product)
    void gram_w_only(const T *Wp, const int *done) {
        EpiStore<T> eg{slabs.p + gram_slab_off, K, (int64_t)K * K, nullptr};
        gemm<KCONTIG, KCONTIG>("gemm_WtW", Wp, P, K, Wp, P, K, P, s_gw, true, eg, done, (double)(P * K) * sizeof(T));
        reduce_slabs_from("reduce_WtW", gramW_p, slabs.p + gram_slab_off, (int64_t)K * K, s_gw, done);
    }
    void gram_h_only(const T *Hp, const int *done) {
        EpiStore<T> eg{slabs.p + gram_slab_off, K, (int64_t)K * K, nullptr};
        gemm<KSTRIDED, KSTRIDED>("gemm_HHt", Hp, K, K, Hp, K, K, N, s_gh, true, eg, done, (double)(K * N) * sizeof(T));
        reduce_slabs_from("reduce_HHt", gramH_p, slabs.p + gram_slab_off, (int64_t)K * K, s_gh, done);
    }
    hipEvent_t ev_red[PIPE_C] = {}, ev_rs[PIPE_C] = {}, ev_pack[PIPE_C] = {}, ev_ag[PIPE_C] = {}, ev_tail = nullptr;
    bool pipe_pending = false;            // an iteration's W is still in flight (all-gather not consumed, stop check not run)
    long long pipe_t = 0;                 // ... that iteration's number
    int pipe_gram_pieces = 1;             // Gram tail pieces per chunk launch
    bool pipelined() const { return row_sharded() && comm_mode == NMFX_COMM_PIPELINED && Pcc > 0 && Pcc % 128 == 0 && fuse_gram && K % 128 == 0 && !use_bf16x3(); }
    void enqueue_multmse_pipelined(const nmfx_opts &o, long long t);
    void pipe_consume_w(const nmfx_opts &o, bool launch_wtx, bool run_check);
    void pipe_flush(const nmfx_opts &o);
    void wt_times_chunk(const T *Wp, const T *Bmat, int c, const int *done);
    void times_ht_chunk(const T *Amat, const T *Hp, int c, const int *done);
    static constexpr int CT = sizeof(T) == 4 ? CT_F32 : CT_F64;

    // profiling (hipEvent pair per launch, resolved lazily)
    struct Rec { const char *name; int ev; double flops, bytes; };
    int profiling = 0;
    std::map<std::string, long long> prof_seen;   // mode 2: launches seen per GEMM name
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    std::vector<Rec> records;
    int ev_used = 0;

    void require_ready() {
        if (!have_X) throw StatusError{NMFX_ERR_STATE, "X has not been uploaded (nmfx_set_X)"};
        if (!have_F) throw StatusError{NMFX_ERR_STATE, "W/H have not been uploaded (nmfx_set_factors)"};
    }

    // grid size of a flat grid-stride elementwise launch (256 threads, >= 4 elements per thread when large)
    unsigned flat_grid(int64_t count) const {
        return (unsigned)std::max<int64_t>(1, std::min<int64_t>((count + 255) / 256, (int64_t)num_cu * 32));
    }

    int pick_splits(int tiles, int64_t kdim) const {
        const int64_t nkt = kdim / BK;
        int want = (2 * num_cu + tiles - 1) / tiles;
        if (want > 64) want = 64;
        int best = 1;
        for (int d = 1; d <= want; ++d)
            if (nkt % d == 0 && nkt / d >= 8) best = d;
        // short pieces (< 64 k-tiles per block: prologue, first-load latency and epilogue are ~10 % of such a block, and every split is a
        // slab to write and to combine): back off to the largest split that still gives every CU a block and 64 k-tiles per block
        // (the 8-rank shard of the headline problem: 16 -> 8 splits, 163 -> 159 us in gemm_bench 6 4, half the slabs to combine)
        if (nkt / best < 64)
            for (int d = best - 1; d >= 1; --d)
                if (nkt % d == 0 && nkt / d >= 64 && (int64_t)tiles * d >= num_cu) { best = d; break; }
        return best;
    }

    template <typename F> void timed(const char *name, double flops, double bytes, F &&launch) {
        if (!profiling) { launch(); return; }
        // mode 2 / 3: only the products that carry the iteration's flops (the p*n*k ones, not the k x k x n Gram / update products)
        // and the collectives of the exchange step (names "comm_..."), every 8th (mode 2) / every 2nd (mode 3) launch of each
        if (profiling >= 2) {
            const bool wanted = flops >= (double)P * (double)N * (double)K || std::strncmp(name, "comm_", 5) == 0;
            if (!wanted || ((prof_seen[name]++) & (profiling == 2 ? 7 : (profiling == 3 ? 3 : 15))) != 0) { launch(); return; }
        }
        if (ev_used == (int)ev_pool.size()) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            ev_pool.emplace_back(a, b);
        }
        const int id = ev_used++;
        HIP_TRY(hipEventRecord(ev_pool[id].first, stream));
        launch();
        HIP_TRY(hipEventRecord(ev_pool[id].second, stream));
        records.push_back(Rec{name, id, flops, bytes});
    }

    template <int LA, int LB, int BR, int BC, int WGR, int WGC, int AUX, typename Epi>
    void launch_gemm_cfg(const GemmArgs<T> &g, const Epi &epi) {
        const int blocks = g.tiles_r * g.tiles_c * g.splits - g.tail_main;
        // operand staging by buffer loads everywhere (gemm_mfma.hpp, BUF); the k-loop unrolled by two on top of it where that measured
        // faster: Float32, both operands contraction-contiguous, 128 x 128 tiles (W'X: 1117 -> 1087 us in gemm_bench 6 3; the strided
        // layout +1 %, Float64 +16 %: left alone)
        // (not under ProjectedALS's factorisations: the Cholesky workgroup that shares a CU with a block of the product lives on the
        // issue slots the product leaves, and the unrolled loop leaves fewer -- potrf 590 -> 915 us co-resident, which put the chain
        // back on the critical path: 2.21 -> 2.30 ms per iteration)
        constexpr int BUFV = (sizeof(T) == 4 && LA == KCONTIG && LB == KCONTIG && BR == 128 && BC == 128 && AUX == 0) ? 2 : 1;
        if (BUFV == 2 && (!short_grid || (chol_unrolled && potrf_reg_ok())))
            hipLaunchKernelGGL((gemm_mfma_kernel<T, LA, LB, BR, BC, WGR, WGC, Epi, AUX, BUFV>), dim3(blocks), dim3(WGR * WGC * 64), 0,
                               stream, g, epi);
        else
            hipLaunchKernelGGL((gemm_mfma_kernel<T, LA, LB, BR, BC, WGR, WGC, Epi, AUX, 1>), dim3(blocks), dim3(WGR * WGC * 64), 0,
                               stream, g, epi);
        HIP_TRY(hipGetLastError());
    }

    // D(R x C) = sum_k A(r,k) B(c,k); R, C, Kdim are padded sizes.  Picks the block tile from R, C.
    struct Seg {   // optional second operand segment (see GemmArgs)
        const T *A2 = nullptr; int64_t lda2 = 0, r_split = INT64_MAX;
        const T *B2 = nullptr; int64_t ldb2 = 0, c_split = INT64_MAX;
        int tail_tiles = 0;    // extra tiles along the slow direction, processed as a balanced tail segment
        int tail_main = 0, tail_per = 0;   // short grid: the last tail_main (tile, split) items as pieces of tail_per k-tiles
        const T *a_aux = nullptr, *b_aux = nullptr;   // operand computed on the fly (projected-gradient trial step)
        const double *alpha_ptr = nullptr;
        int64_t b_blk_k = 0, b_blk_stride = 0;        // B operand blocked along the contraction (GemmArgs)
        const int *sel = nullptr;                     // operands in device-indexed buffer sets (GemmArgs::sel)
        int64_t a_sel = 0, b_sel = 0, aaux_sel = 0, baux_sel = 0;
    };
    template <int LA, int LB, int AUX = 0, typename Epi>
    void gemm(const char *name, const T *A, int64_t lda, int64_t R, const T *B, int64_t ldb, int64_t C, int64_t Kdim,
              int splits, bool c_fastest, const Epi &epi, const int *done, double bytes = 0.0, const Seg &seg = Seg(),
              double extra_flops = 0.0) {
        GemmArgs<T> g;
        g.A = A; g.B = B; g.lda = lda; g.ldb = ldb;
        g.A2 = seg.A2; g.lda2 = seg.lda2; g.r_split = seg.r_split;
        g.B2 = seg.B2; g.ldb2 = seg.ldb2; g.c_split = seg.c_split;
        g.tail_tiles = seg.tail_tiles; g.tail_nkt = (int)(Kdim / BK);
        g.tail_main = seg.tail_main; g.tail_per = seg.tail_per;
        g.a_aux = seg.a_aux; g.b_aux = seg.b_aux; g.alpha_ptr = seg.alpha_ptr;
        g.b_blk_k = seg.b_blk_k; g.b_blk_stride = seg.b_blk_stride;
        g.sel = seg.sel; g.a_sel = seg.a_sel; g.b_sel = seg.b_sel; g.aaux_sel = seg.aaux_sel; g.baux_sel = seg.baux_sel;
        g.splits = splits;
        g.kchunk = (int)(Kdim / splits);
        g.c_fastest = c_fastest ? 1 : 0;
        g.done = done;
        const double flops = 2.0 * (double)R * (double)C * (double)Kdim + extra_flops;
        auto note = [&] { last_tiles_r = g.tiles_r; last_blocks = g.tiles_r * g.tiles_c * g.splits; };
        // Short contractions (the k x k Gram products: 8 k-tiles) are dominated by prologue/epilogue latency; give
        // them half-size tiles so >= 2 blocks per CU are resident and one block's epilogue overlaps another's MFMAs.
        // Epilogues whose 128 x 128 instantiation is left at ONE wave per SIMD always run on half-size tiles (2-3 waves):
        // f64 with anything but a plain store (128 accumulator registers per lane), and the update / gradient / line-search
        // epilogues in f32 (Epi::HEAVY; see scripts/kernel_regs.py).
        constexpr bool prefer_half = ((sizeof(T) == 8) && !std::is_same<Epi, EpiStore<T>>::value) || Epi::HEAVY;
        const bool small_k = (Kdim <= 1024) && splits == 1 && seg.tail_tiles == 0 &&
                             (prefer_half || (R / 128) * (C / 128) < 2 * (int64_t)num_cu);
        timed(name, flops, bytes, [&] {
            // heavy epilogues (update / gradient / line search) also when the half-size tiles make exactly ONE resident wave of two
            // blocks per CU: those run prologue, 16 k-tiles and epilogue in lockstep; four quarter-size blocks per CU overlap one
            // block's epilogue with another's k-loop (update products at 16384 x 256 x 256: 45 -> 38 us)
            const int64_t half_blocks = std::max(R / 64 * (C / 128), R / 128 * (C / 64));
            if (force_quarter_tiles && R % 64 == 0 && C % 64 == 0 && seg.tail_tiles == 0) {
                // a small split-K product that would put only a few dozen 128 x 128 blocks on the chip (the own-rows Gram of the
                // row-sharded W side): 4x the blocks on quarter-size tiles
                g.tiles_r = (int)(R / 64); g.tiles_c = (int)(C / 64);
                launch_gemm_cfg<LA, LB, 64, 64, 2, 2, AUX>(g, epi);
            } else if (small_k && R % 64 == 0 && C % 64 == 0 && (half_blocks < 2 * (int64_t)num_cu || (Epi::HEAVY && half_blocks == 2 * (int64_t)num_cu))) {
                // even the half-size tiles leave CUs idle (e.g. the 4096 x 512 projected-gradient products of a C5 shard:
                // 256 blocks): quarter-size tiles, 4 waves of 32 x 32
                g.tiles_r = (int)(R / 64); g.tiles_c = (int)(C / 64);
                launch_gemm_cfg<LA, LB, 64, 64, 2, 2, AUX>(g, epi);
            } else if (small_k && R % 64 == 0 && C % 128 == 0 && R >= C) {
                g.tiles_r = (int)(R / 64); g.tiles_c = (int)(C / 128);
                launch_gemm_cfg<LA, LB, 64, 128, 1, 4, AUX>(g, epi);
            } else if (small_k && R % 128 == 0 && C % 64 == 0 && C > R) {
                g.tiles_r = (int)(R / 128); g.tiles_c = (int)(C / 64);
                launch_gemm_cfg<LA, LB, 128, 64, 4, 1, AUX>(g, epi);
            } else if (small_k && (C == 64 || R == 64) && R % 64 == 0 && C % 64 == 0) {
                // k <= 64: 64 x 64 tiles give 4x the blocks of the 256 x 64 / 64 x 256 shapes (16 -> 64 at C2)
                g.tiles_r = (int)(R / 64); g.tiles_c = (int)(C / 64);
                launch_gemm_cfg<LA, LB, 64, 64, 2, 2, AUX>(g, epi);
            } else if (R % 128 == 0 && C % 128 == 0) {
                g.tiles_r = (int)(R / 128); g.tiles_c = (int)(C / 128);
                if (splits == 1 && g.tiles_r >= 16 && g.tiles_c >= 16 && g.tiles_r % 8 == 0 && g.tiles_c % 8 == 0) g.group = 8;
                if (g.tail_tiles > 0) g.tail_per = tail_piece(g.tiles_r * g.tiles_c * splits, g.tail_tiles * (c_fastest ? g.tiles_c : g.tiles_r), g.tail_nkt);
                launch_gemm_cfg<LA, LB, 128, 128, 2, 2, AUX>(g, epi);
            } else if (C == 64 && R % 256 == 0) {
                g.tiles_r = (int)(R / 256); g.tiles_c = 1;
                launch_gemm_cfg<LA, LB, 256, 64, 4, 1, AUX>(g, epi);
            } else if (R == 64 && C % 256 == 0) {
                g.tiles_r = 1; g.tiles_c = (int)(C / 256);
                launch_gemm_cfg<LA, LB, 64, 256, 1, 4, AUX>(g, epi);
            } else if (R == 64 && C == 64) {
                g.tiles_r = 1; g.tiles_c = 1;
                launch_gemm_cfg<LA, LB, 64, 64, 2, 2, AUX>(g, epi);
            } else if (R % 128 == 0 && C % 64 == 0 && seg.tail_tiles == 0) {
                // one output dimension is a multiple of 64 only (K = 192, 320, ...: the 64-granular component padding): half-width tiles
                g.tiles_r = (int)(R / 128); g.tiles_c = (int)(C / 64);
                launch_gemm_cfg<LA, LB, 128, 64, 4, 1, AUX>(g, epi);
            } else if (R % 64 == 0 && C % 128 == 0 && seg.tail_tiles == 0) {
                g.tiles_r = (int)(R / 64); g.tiles_c = (int)(C / 128);
                launch_gemm_cfg<LA, LB, 64, 128, 1, 4, AUX>(g, epi);
            } else if (R % 64 == 0 && C % 64 == 0 && seg.tail_tiles == 0) {
                g.tiles_r = (int)(R / 64); g.tiles_c = (int)(C / 64);
                launch_gemm_cfg<LA, LB, 64, 64, 2, 2, AUX>(g, epi);
            } else {
                throw StatusError{NMFX_ERR_UNSUPPORTED, "internal: no tile configuration for this GEMM shape"};
            }
        });
        note();
    }

    static bool vec16_ok(const void *a, const void *b, const void *c) {
        return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
    }
    void reduce_slabs_from(const char *name, T *dst, const T *src, int64_t count, int nslab, const int *done,
                           int64_t stride = -1) {
        if (stride < 0) stride = count;
        timed(name, 0.0, (double)count * (nslab + 1) * sizeof(T), [&] {
            const int bs = 256;
            if (nslab >= 16)
                hipLaunchKernelGGL(reduce_many_slabs_kernel<T>, dim3((unsigned)((count + 63) / 64)), dim3(256), 0, stream, dst,
                                   src, count, nslab, stride, done);
            else if (constexpr int V = 16 / (int)sizeof(T); count % V == 0 && stride % V == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0 &&
                                                        (reinterpret_cast<uintptr_t>(src) & 15) == 0)
                hipLaunchKernelGGL(reduce_slabs_vec_kernel<T>, dim3((unsigned)((count / V + bs - 1) / bs)), dim3(bs), 0, stream, dst,
                                   src, count / V, nslab, stride, done);
            else
                hipLaunchKernelGGL(reduce_slabs_kernel<T>, dim3((unsigned)((count + bs - 1) / bs)), dim3(bs), 0, stream, dst,
                                   src, count, nslab, stride, done);
            HIP_TRY(hipGetLastError());
        });
    }
    // numerator and Gram of one side combined by ONE launch
    void reduce_pair(const char *name, T *dst1, const T *src1, int64_t count1, int nslab1, int64_t stride1, T *dst2, const T *src2,
                     int64_t count2, int nslab2, int64_t stride2, const int *done) {
        timed(name, 0.0, ((double)count1 * (nslab1 + 1) + (double)count2 * (nslab2 + 1)) * sizeof(T), [&] {
            const unsigned nb1 = (unsigned)((count1 + 63) / 64), nb2 = (unsigned)((count2 + 63) / 64);
            hipLaunchKernelGGL(reduce_pair_kernel<T>, dim3(nb1 + nb2), dim3(256), 0, stream, dst1, src1, count1, nslab1, stride1, dst2, src2,
                               count2, nslab2, stride2, nb1, done);
            HIP_TRY(hipGetLastError());
        });
    }
    void reduce_slabs(const char *name, T *dst, int64_t count, int nslab, const int *done) {
        reduce_slabs_from(name, dst, slabs.p, count, nslab, done);
    }

    // ---- shared building blocks ------------------------------------------------
    void reduce_to(const char *name, T *dst, int64_t count, int nslab, const int *done) { reduce_slabs(name, dst, count, nslab, done); }

    // k-tiles per tail piece: smallest piece such that (tail tiles) x (pieces per tile) <= blocks, and the pieces fit
    // the Gram slab buffer.  One piece per block => every block does kchunk/BK + tail_per k-tiles.
    int tail_piece(int blocks, int tail_tiles_total, int nkt) const {
        int per = std::max(1, (int)(((int64_t)tail_tiles_total * nkt + blocks - 1) / blocks));
        while ((int64_t)tail_tiles_total * ((nkt + per - 1) / per) > blocks || (nkt + per - 1) / per > max_gram_slabs / PIPE_C) ++per;
        return per;
    }

    // numH = W' * Bmat  (K x N, ld K), Bmat = X or Q (P x N)      src/multupd.jl:98,175; projals.jl:93; alspgrad.jl:66
    // with_gram: also gramW = W'W (src/projals.jl:92, alspgrad.jl:65) in the SAME launch: the A operand is the
    // row-concatenation [Bmat ; W], the output the K x (N+K) matrix [numH | gramW] (hside).
    // keep_slabs: leave the result as split-K slabs in the H region (caller consumes h_nslab slabs of stride h_stride).
    int h_nslab = 1;
    int64_t h_stride = 0;
    // nmfx_opts.precision = NMFX_PREC_BF16X3: the two p*n*k products run on the bf16 matrix cores (gemm_bf16x3.hpp)
    int precision = 0;
    bool use_bf16x3() const { return precision == 1 && sizeof(T) == 4 && K % 128 == 0; }
    template <int LA, int LB, typename Epi>
    void launch_bf16x3(const char *name, const T *A, int64_t lda, int64_t R, const T *B, int64_t ldb, int64_t C, int64_t Kdim, int splits,
                       bool c_fastest, const Epi &epi, const int *done, double bytes) {
        if constexpr (sizeof(T) == 4) {
            bf16x3::Args g;
            g.A = reinterpret_cast<const float *>(A); g.B = reinterpret_cast<const float *>(B);
            g.lda = lda; g.ldb = ldb;
            g.tiles_r = (int)(R / 128); g.tiles_c = (int)(C / 128); g.splits = splits; g.kchunk = (int)(Kdim / splits);
            g.c_fastest = c_fastest ? 1 : 0;
            g.group = (splits == 1 && g.tiles_r >= 16 && g.tiles_c >= 16 && g.tiles_r % 8 == 0 && g.tiles_c % 8 == 0) ? 8 : 1;
            g.done = done;
            const int blocks = g.tiles_r * g.tiles_c * splits;
            timed(name, 2.0 * (double)R * (double)C * (double)Kdim, bytes, [&] {
                hipLaunchKernelGGL((bf16x3::gemm_bf16x3_kernel<LA, LB, Epi>), dim3((unsigned)blocks), dim3(256), 0, stream, g, epi);
                HIP_TRY(hipGetLastError());
            });
            last_tiles_r = g.tiles_r; last_blocks = blocks;
        }
    }
    // the W*H products with a fused epilogue (ratio pass, objective): fp32 kernel, or the bf16x3 one when opted in
    // f32, K a multiple of 128, an output of many tiles per block slot: the persistent cross-tile pipeline of gemm_stream.hpp
    // (tile i's epilogue runs under tile i + 1's MFMAs).  NMFX_STREAM_WH=0 keeps the block-per-tile kernel.
    bool stream_wh = true;
    template <int FAST> static SEpiRatio<float, FAST> to_stream(const EpiRatio<float, FAST> &e) { return SEpiRatio<float, FAST>{e.X, e.Q, e.ld, e.delta}; }
    template <int KL> static SEpiObjective<float, KL> to_stream(const EpiObjective<float, KL> &e) { return SEpiObjective<float, KL>{e.X, e.ld, e.partial, 0.0}; }
    template <typename Epi>
    void gemm_wh(const char *name, const T *Hp, const T *Wp, const Epi &epi, const int *done, double bytes) {
        if (use_bf16x3() && P % 128 == 0 && N % 128 == 0) { launch_bf16x3<0, 1>(name, Hp, K, N, Wp, P, P, K, 1, true, epi, done, bytes); return; }
        if constexpr (sizeof(T) == 4) {
            const int64_t tiles = (N / 128) * (P / 128);
            const int grid = 2 * num_cu;
            if (stream_wh && K % 128 == 0 && (grid & 7) == 0 && tiles >= 4 * (int64_t)grid && tiles < (int64_t)1 << 30) {
                StreamArgs g;
                g.A = reinterpret_cast<const float *>(Hp); g.lda = K;
                g.B = reinterpret_cast<const float *>(Wp); g.ldb = P;
                g.tiles_r = (int)(N / 128); g.tiles_c = (int)(P / 128);
                g.nkt = (int)(K / 32);
                g.group = (g.tiles_r % 8 == 0 && g.tiles_c % 8 == 0) ? 8 : 1;
                g.done = done;
                auto se = to_stream(epi);
                timed(name, 2.0 * (double)N * (double)P * (double)K, bytes, [&] {
                    hipLaunchKernelGGL((gemm_wh_stream_kernel<decltype(se)>), dim3((unsigned)grid), dim3(256), 0, stream, g, se);
                    HIP_TRY(hipGetLastError());
                });
                last_tiles_r = g.tiles_r; last_blocks = grid;
                return;
            }
        }
        gemm<KCONTIG, KSTRIDED>(name, Hp, K, N, Wp, P, P, K, 1, true, epi, done, bytes);
    }
    // A product whose (tile, split) items would fill every block slot of the chip (two 128 x 128 blocks per CU, all resident for
    // the whole launch) while another stream has a workgroup to place: launch `chol_slots` blocks short and deal the missing items
    // out as tail pieces (GemmArgs::tail_main), +chol_slots / (2 num_cu) of work per block.  leftover = 0: launch as usual;
    // `inner` = tiles along the fast tile direction (leftover tiles are whole lines of it: a rectangle of the output).
    struct ShortGrid { int leftover = 0, per = 0, pieces = 0; };
    ShortGrid plan_short_grid(int tiles, int splits, int inner, int64_t Kdim) const {
        ShortGrid sg;
        if (!short_grid || chol_slots <= 0 || K % 128 != 0) return sg;
        const int items = tiles * splits, slots = 2 * num_cu - chol_slots;
        if (items <= slots || items > 2 * num_cu) return sg;      // already leaves room / more than one wave of blocks
        int left = items - slots;
        left = (left + inner - 1) / inner * inner;
        while (((items - left) & 7) && left < tiles) left += inner;
        const int grid = items - left;
        if (left > tiles || grid <= 0 || (grid & 7)) return sg;
        const int nkt = (int)(Kdim / splits / BK);
        int per = std::max(1, (int)(((int64_t)nkt * left + grid - 1) / grid));
        while ((int64_t)left * ((nkt + per - 1) / per) > grid) ++per;
        const int pieces = (nkt + per - 1) / per;
        if ((size_t)pieces * left * 128 * 128 > (size_t)max_gram_slabs * K * K) return sg;
        sg.leftover = left; sg.per = per; sg.pieces = pieces;
        return sg;
    }
    void reduce_pieces(const char *name, T *dst, int64_t ldd, const T *src, int64_t rows, int64_t cols, int npieces, const int *done) {
        timed(name, 0.0, (double)rows * cols * (npieces + 1) * sizeof(T), [&] {
            hipLaunchKernelGGL(reduce_pieces_kernel<T>, dim3((unsigned)std::min<int64_t>((rows * cols + 255) / 256, 4096)), dim3(256), 0, stream, dst, ldd, src,
                               rows, cols, npieces, rows * cols, done);
            HIP_TRY(hipGetLastError());
        });
    }
    void wt_times(const T *Wp, const T *Bmat, bool with_gram, const int *done, bool keep_slabs = false) {
        T *reg = slabs.p;
        if (use_bf16x3()) {
            h_nslab = s_h; h_stride = (int64_t)K * N;
            {
                EpiStore<T> e{reg, K, h_stride, nullptr};
                launch_bf16x3<0, 0>("gemm_WtX_bf16x3", Bmat, P, N, Wp, P, K, P, s_h, true, e, done, (double)(P * N + P * K) * sizeof(T));
            }
            if (!keep_slabs || h_nslab > 2) { reduce_slabs_from("reduce_WtX", numH_p, reg, h_stride, h_nslab, done); h_in_slabs = false; }
            else h_in_slabs = true;
            if (with_gram) {   // the k x k Gram stays on the fp32 path
                EpiStore<T> eg{slabs.p + gram_slab_off, K, (int64_t)K * K, nullptr};
                gemm<KCONTIG, KCONTIG>("gemm_WtW", Wp, P, K, Wp, P, K, P, s_gw, true, eg, done, (double)(P * K) * sizeof(T));
                reduce_slabs_from("reduce_WtW", gramW_p, slabs.p + gram_slab_off, (int64_t)K * K, s_gw, done);
            }
            return;
        }
        if (with_gram && fuse_gram && K % 128 == 0) {
            h_nslab = s_h; h_stride = (int64_t)K * N;
            const int tiles = (int)((N / 128) * (K / 128));
            const int tt = (int)((K / 128) * (K / 128));
            const int per = tail_piece(tiles * s_h, tt, (int)(P / BK));
            const int pieces = (int)((P / BK + per - 1) / per);
            // (an unsplit product whose consumer wants the combined numerator stores it where the combine would have copied it)
            const bool straight = s_h == 1 && !keep_slabs;
            EpiStore<T> e{straight ? numH_p : reg, K, h_stride, nullptr};
            e.C2 = slabs.p + gram_slab_off; e.ld2 = K; e.stride2 = (int64_t)K * K; e.r_off = N; e.c_off = 0;
            Seg sg;
            sg.A2 = Wp; sg.lda2 = P; sg.r_split = N; sg.tail_tiles = (int)(K / 128);
            gemm<KCONTIG, KCONTIG>("gemm_WtX", Bmat, P, N, Wp, P, K, P, s_h, true, e, done,
                                   (double)(P * N + 2 * P * K) * sizeof(T), sg, 2.0 * K * K * P);
            if (straight) {
                reduce_slabs_from("reduce_WtW", gramW_p, slabs.p + gram_slab_off, (int64_t)K * K, pieces, done);
                h_in_slabs = false;
                return;
            }
            if (h_reduce_pair && (!keep_slabs || h_nslab > 2)) {   // both combines in one launch
                reduce_pair("reduce_WtX_WtW", numH_p, reg, h_stride, h_nslab, h_stride, gramW_p, slabs.p + gram_slab_off, (int64_t)K * K, pieces,
                            (int64_t)K * K, done);
                h_in_slabs = false;
                return;
            }
            reduce_slabs_from("reduce_WtW", gramW_p, slabs.p + gram_slab_off, (int64_t)K * K, pieces, done);
            if (!keep_slabs || h_nslab > 2) {
                reduce_slabs_from("reduce_WtX", numH_p, reg, h_stride, h_nslab, done);
                h_in_slabs = false;
            } else {
                h_in_slabs = true;
            }
            return;
        }
        h_nslab = s_h; h_stride = (int64_t)K * N;
        EpiStore<T> e{reg, K, h_stride, nullptr};
        const ShortGrid shg = with_gram ? ShortGrid() : plan_short_grid((int)((N / 128) * (K / 128)), s_h, (int)(K / 128), P);
        if (shg.leftover) {
            // the last lines of tiles (columns of numH from c0 on) of the last split arrive as pieces
            const int64_t lines = shg.leftover / (K / 128), c0 = N - lines * 128;
            e.C2 = slabs.p + gram_slab_off; e.ld2 = K; e.stride2 = lines * 128 * K; e.r_off = c0; e.c_off = 0;
            Seg sg;
            sg.tail_main = shg.leftover; sg.tail_per = shg.per;
            gemm<KCONTIG, KCONTIG>("gemm_WtX", Bmat, P, N, Wp, P, K, P, s_h, true, e, done, (double)(P * N + P * K) * sizeof(T), sg);
            if (!keep_slabs || h_nslab > 2) {   // pieces + slabs in one launch
                timed("reduce_WtX", 0.0, (double)K * N * (h_nslab + 1) * sizeof(T), [&] {
                    // (every size here is a multiple of 64; the buffers are hipMalloc'ed and offset by multiples of 64 elements)
                    constexpr int V = 16 / (int)sizeof(T);
                    if (vec16_ok(numH_p, reg, slabs.p + gram_slab_off))
                        hipLaunchKernelGGL((reduce_slabs_tail_kernel<T, V>), dim3((unsigned)((K * N / V + 255) / 256)), dim3(256), 0, stream, numH_p, reg, (int64_t)K * N,
                                           h_nslab, h_stride, slabs.p + gram_slab_off, shg.pieces, lines * 128 * K, 0, c0 * K, lines * 128 * K, (int64_t)0,
                                           (int64_t)0, (int64_t)0, done);
                    else
                        hipLaunchKernelGGL((reduce_slabs_tail_kernel<T, 1>), dim3((unsigned)((K * N + 255) / 256)), dim3(256), 0, stream, numH_p, reg, (int64_t)K * N,
                                           h_nslab, h_stride, slabs.p + gram_slab_off, shg.pieces, lines * 128 * K, 0, c0 * K, lines * 128 * K, (int64_t)0,
                                           (int64_t)0, (int64_t)0, done);
                    HIP_TRY(hipGetLastError());
                });
                h_in_slabs = false;
                return;
            }
            reduce_pieces("reduce_WtX_pieces", reg + (int64_t)(s_h - 1) * h_stride + c0 * K, lines * 128 * K, slabs.p + gram_slab_off, 1, lines * 128 * K,
                          shg.pieces, done);
        } else if (wt_blocked != nullptr) {
            // W as the all-gather left it: rank q's Pc rows x K at wt_blocked + q * wt_blk_stride (ld Pc); split s contracts over rows of ONE block
            Seg sg;
            sg.b_blk_k = Pc; sg.b_blk_stride = wt_blk_stride;
            gemm<KCONTIG, KCONTIG>("gemm_WtX", Bmat, P, N, wt_blocked, Pc, K, P, s_h, true, e, done, (double)(P * N + P * K) * sizeof(T), sg);
        } else
        gemm<KCONTIG, KCONTIG>("gemm_WtX", Bmat, P, N, Wp, P, K, P, s_h, true, e, done,
                               (double)(P * N + P * K) * sizeof(T));
        const bool red = !keep_slabs || h_nslab > h_keep_max;
        h_in_slabs = !red;
        if (with_gram) {
            EpiStore<T> eg{slabs.p + gram_slab_off, K, (int64_t)K * K, nullptr};
            gemm<KCONTIG, KCONTIG>("gemm_WtW", Wp, P, K, Wp, P, K, P, s_gw, true, eg, done, (double)(P * K) * sizeof(T));
            if (red) reduce_pair("reduce_WtX_WtW", numH_p, reg, h_stride, h_nslab, h_stride, gramW_p, slabs.p + gram_slab_off, (int64_t)K * K, s_gw,
                                 (int64_t)K * K, done);
            else reduce_slabs_from("reduce_WtW", gramW_p, slabs.p + gram_slab_off, (int64_t)K * K, s_gw, done);
        } else if (red) {
            reduce_slabs_from("reduce_WtX", numH_p, reg, h_stride, h_nslab, done);
        }
    }
    const T *wt_blocked = nullptr;   // set around wt_times: W in the blocked layout of the all-gather (plain branch only)
    int64_t wt_blk_stride = 0;
    // Blocked residency of W (row-sharded fused MultUpdate-MSE step, transports whose all-gather lands in local memory): between
    // iterations W lives as the all-gather delivers it -- rank q's rows at Wblk[wb] + q * blk_chunk, followed by the rank's
    // stop_condition partials -- and is only unpacked into the standard layout when something else needs it (w_sync()).
    DevBuf<unsigned char> Wblk[2];
    int wb = 0, blk_cpp = 1;
    size_t blk_chunk = 0;
    bool w_res_blocked = false, w_std_stale = false;
    bool blk_enabled = true;         // NMFX_W_BLOCKED=0: unpack after every all-gather (A/B)
    bool peer_pull_enabled = true;   // NMFX_P2P_PULL=0: the peer transport's second exchange as a push + unpack (round 5; A/B)
    bool blocked_residency_ok() const {
        if (!(blk_enabled && rs_fused() && s_h % nranks == 0 && Pc % (P / s_h) == 0 && !short_grid)) return false;
        // peer transport: the chunk and the own-rows Gram must fit one slot of the window (the pull form of the second exchange)
        if (PeerComm *pc = peer()) {
            const size_t chunk = ((size_t)Pc * K * sizeof(T) + (size_t)64 * 2 * K * sizeof(double) + 255) / 256 * 256;
            return peer_pull_enabled && nranks <= EPI_MAX_PIECES && chunk + (size_t)K * K * sizeof(T) + 512 <= pc->slot_bytes;
        }
        return true;
    }
    void w_sync(const int *done) {   // W[wcur] <- the blocked copy
        if (!w_res_blocked || !w_std_stale) return;
        hipLaunchKernelGGL(gathered_to_full_kernel<T>, dim3(flat_grid(P * K)), dim3(256), 0, stream, W[wcur].p, Wblk[wb].p, nranks, blk_chunk, P, K, Pc,
                           (int64_t)0, 0, (double *)nullptr, 0, done);
        HIP_TRY(hipGetLastError());
        w_std_stale = false;
    }
    // after wt_times(..., with_gram=true): where the numerator / the Gram operand live
    bool h_in_slabs = false, w_in_slabs = false;
    const T *h_num() const { return h_in_slabs ? slabs.p : numH_p; }
    int h_keep_max = 2;   // slabs the consumer of wt_times(keep_slabs) can sum itself (EpiMultUpdate<.., NSL>: 2, or 8 in the row-sharded fused step)
    int h_num_nslab() const {
        if (h_in_slabs && h_nslab > h_keep_max) throw StatusError{NMFX_ERR_UNSUPPORTED, "internal: more numerator slabs left than the update epilogue sums"};
        return h_in_slabs ? h_nslab : 1;
    }

    // numW = Amat * H'  (P x K, ld P), Amat = X or Q              src/multupd.jl:109,187; projals.jl:101; alspgrad.jl:221
    // with_gram: also gramH = HH' (src/projals.jl:100, alspgrad.jl:220) in the same launch: B operand = [Amat ; H],
    // slab = [ numW (ld P) | gramH (ld K) ] = the layout of the packed all-reduce buffer.
    int w_nslab = 1;
    int64_t w_stride = 0;
    void times_ht(const T *Amat, const T *Hp, bool with_gram, const int *done, bool keep_slabs = false, const T *HtP = nullptr) {
        T *reg = slabs.p + slab_w_off;
        // HtP = H' (N x K, ld N) of the same H: the product contracts over the contiguous index of X' and H' (see Xt above)
        const bool xt = HtP != nullptr && Amat == X.p && xt_valid && !use_bf16x3();
        auto big = [&](int splits, const auto &e, double bytes, const Seg &sg_in, double extra) {
            if (xt) {
                Seg sg = sg_in;
                if (sg.B2 != nullptr) { sg.B2 = HtP; sg.ldb2 = N; }
                gemm<KCONTIG, KCONTIG>("gemm_XHt", HtP, N, K, Xt.p, N, P, N, splits, false, e, done, bytes, sg, extra);
            } else {
                gemm<KSTRIDED, KSTRIDED>("gemm_XHt", Hp, K, K, Amat, P, P, N, splits, false, e, done, bytes, sg_in, extra);
            }
        };
        if (use_bf16x3()) {
            w_nslab = s_w; w_stride = (int64_t)P * K;
            {
                EpiStore<T> e{reg, P, w_stride, nullptr};
                launch_bf16x3<1, 1>("gemm_XHt_bf16x3", Hp, K, K, Amat, P, P, N, s_w, false, e, done, (double)(P * N + K * N) * sizeof(T));
            }
            finish_w_slabs(keep_slabs, done);
            if (with_gram) {
                EpiStore<T> eg{slabs.p + gram_slab_off, K, (int64_t)K * K, nullptr};
                gemm<KSTRIDED, KSTRIDED>("gemm_HHt", Hp, K, K, Hp, K, K, N, s_gh, true, eg, done, (double)(K * N) * sizeof(T));
                reduce_slabs_from("reduce_HHt", gramH_p, slabs.p + gram_slab_off, (int64_t)K * K, s_gh, done);
            }
            return;
        }
        if (with_gram && fuse_gram && K % 128 == 0) {
            // row-sharded fused step at a short local contraction: ONE split, the product stored straight into the blocked send
            // buffer (EpiStore::piece_rows) -- no slabs to write and to combine (measured at the 8-rank shard shape of the headline
            // problem, scripts/kbench/gemm_bench.hip: 153 us unsplit on 256 blocks against 150 us 2-way split on 512)
            // (only where it pays: a local contraction of 1024 .. 4096 -- 4 or more ranks at the headline shape -- and enough tiles to
            // give every CU one; longer contractions keep the 2-way split that puts two blocks on a CU)
            // (round 6: no upper limit on the send-buffer route any more -- one block per CU keeps the matrix pipes as busy as two at a long
            // contraction too (solver_impl.hpp: iterate), and there is no slab to combine: the 2-rank shard of the headline problem, 256
            // k-tiles, 1.030 -> 1.008 ms per simulated rank; the peer transport's epilogue, whose blocks store a whole tile into uncached
            // peer memory, measured SLOWER there, 1.049 -> 1.077 ms, and keeps the limit: profiles/r06_simranks2_unsplit_send_buffer_ab.jsonl)
            const bool direct = w_defer_combine && w_blocked && N / BK >= 32 && (N / BK <= direct_max_ktiles || (direct_long_local && peer_dst == nullptr)) &&
                                (K / 128) * (P / 128) >= num_cu;
            const int sw = direct ? 1 : s_w;
            w_nslab = sw; w_stride = (int64_t)P * K;
            const int tiles = (int)((K / 128) * (P / 128));
            const int tt = (int)((K / 128) * (K / 128));
            const int per = tail_piece(tiles * sw, tt, (int)(N / BK));
            const int pieces = (int)((N / BK + per - 1) / per);
            Seg sg;
            sg.B2 = Hp; sg.ldb2 = K; sg.c_split = P; sg.tail_tiles = (int)(K / 128);
            if (direct && peer_dst != nullptr) {
                // peer-to-peer transport: row block g of the product goes straight into rank g's receive slot (EpiStorePeer)
                EpiStorePeer<T> e;
                for (int g = 0; g < EPI_MAX_PIECES; ++g) e.piece[g] = (g < nranks) ? peer_dst->num[g] : nullptr;
                e.piece_rows = Pc;
                e.C2 = slabs.p + gram_slab_off; e.ld2 = K; e.stride2 = (int64_t)K * K; e.r_off = 0; e.c_off = P;
                if (const char *ev = dev_env("NMFX_P2P_STAGE16")) e.stage16 = std::atoi(ev) != 0;   // 16-byte staged peer stores (gemm_mfma.hpp: measured slower on the stand-in)
                big(sw, e, (double)(P * N + 2 * K * N) * sizeof(T), sg, 2.0 * K * K * N);
            } else {
            // (straight: an unsplit product whose consumer wants the combined numerator in the standard layout stores it there itself)
            const bool straight = !direct && sw == 1 && !keep_slabs && !w_blocked && !w_defer_combine;
            EpiStore<T> e{(direct || straight) ? numW_p : reg, direct ? Pc : P, w_stride, nullptr};
            if (direct) { e.piece_rows = Pc; e.piece_stride = (int64_t)K * Pc; }
            e.C2 = slabs.p + gram_slab_off; e.ld2 = K; e.stride2 = (int64_t)K * K; e.r_off = 0; e.c_off = P;
            big(sw, e, (double)(P * N + 2 * K * N) * sizeof(T), sg, 2.0 * K * K * N);
            if (straight) {
                w_direct = false;
                reduce_slabs_from("reduce_HHt", gramH_p, slabs.p + gram_slab_off, (int64_t)K * K, pieces, done);
                w_in_slabs = false;
                return;
            }
            }
            w_direct = direct;
            if (w_defer_combine) { w_pieces = pieces; w_in_slabs = false; return; }   // the caller's combine launch sums both
            reduce_slabs_from("reduce_HHt", gramH_p, slabs.p + gram_slab_off, (int64_t)K * K, pieces, done);
            finish_w_slabs(keep_slabs, done);
            return;
        }
        w_nslab = s_w; w_stride = (int64_t)P * K;
        EpiStore<T> e{reg, P, w_stride, nullptr};
        const ShortGrid shg = with_gram ? ShortGrid() : plan_short_grid((int)((K / 128) * (P / 128)), s_w, (int)(K / 128), N);
        if (shg.leftover) {
            // the last lines of tiles (rows of numW from r0 on) of the last split arrive as pieces, compact (ld = rows)
            const int64_t lines = shg.leftover / (K / 128), r0 = P - lines * 128, rows = lines * 128;
            e.C2 = slabs.p + gram_slab_off; e.ld2 = rows; e.stride2 = rows * K; e.r_off = 0; e.c_off = r0;
            Seg sg;
            sg.tail_main = shg.leftover; sg.tail_per = shg.per;
            if (xt) gemm<KCONTIG, KCONTIG>("gemm_XHt", HtP, N, K, Xt.p, N, P, N, s_w, false, e, done, (double)(P * N + K * N) * sizeof(T), sg);
            else gemm<KSTRIDED, KSTRIDED>("gemm_XHt", Hp, K, K, Amat, P, P, N, s_w, false, e, done, (double)(P * N + K * N) * sizeof(T), sg);
            if (!w_blocked && (!keep_slabs || w_nslab > 2)) {   // pieces + slabs in one launch
                timed("reduce_XHt", 0.0, (double)P * K * (w_nslab + 1) * sizeof(T), [&] {
                    constexpr int V = 16 / (int)sizeof(T);
                    if (P % (256 * V) == 0 && vec16_ok(numW_p, reg, slabs.p + gram_slab_off))
                        hipLaunchKernelGGL((reduce_slabs_tail_kernel<T, V>), dim3((unsigned)(P / (256 * V) * K)), dim3(256), 0, stream, numW_p, reg, (int64_t)P * K, w_nslab,
                                           w_stride, slabs.p + gram_slab_off, shg.pieces, rows * K, 1, (int64_t)0, (int64_t)0, P, r0, rows, done);
                    else
                        hipLaunchKernelGGL((reduce_slabs_tail_kernel<T, 1>), dim3((unsigned)(P / 256 * K)), dim3(256), 0, stream, numW_p, reg, (int64_t)P * K, w_nslab,
                                           w_stride, slabs.p + gram_slab_off, shg.pieces, rows * K, 1, (int64_t)0, (int64_t)0, P, r0, rows, done);
                    HIP_TRY(hipGetLastError());
                });
                w_in_slabs = false;
                return;
            }
            reduce_pieces("reduce_XHt_pieces", reg + (int64_t)(s_w - 1) * w_stride + r0, P, slabs.p + gram_slab_off, K, rows, shg.pieces, done);
        } else
        big(s_w, e, (double)(P * N + K * N) * sizeof(T), Seg(), 0.0);
        const bool pair = with_gram && !w_blocked && (!keep_slabs || w_nslab > 2);   // both combines in one launch
        if (!pair) finish_w_slabs(keep_slabs, done);
        if (with_gram) {
            EpiStore<T> eg{slabs.p + gram_slab_off, K, (int64_t)K * K, nullptr};
            gemm<KSTRIDED, KSTRIDED>("gemm_HHt", Hp, K, K, Hp, K, K, N, s_gh, true, eg, done, (double)(K * N) * sizeof(T));
            if (pair) {
                reduce_pair("reduce_XHt_HHt", numW_p, reg, w_stride, w_nslab, w_stride, gramH_p, slabs.p + gram_slab_off, (int64_t)K * K, s_gh,
                            (int64_t)K * K, done);
                w_in_slabs = false;
            } else {
                reduce_slabs_from("reduce_HHt", gramH_p, slabs.p + gram_slab_off, (int64_t)K * K, s_gh, done);
            }
        }
    }
    // where the split-K slabs of X*H' go: left in place for the update GEMM's epilogue (single GPU, <= 2 slabs), summed
    // into numW (standard layout), or summed into the BLOCKED reduce-scatter send buffer (row-sharded W side)
    bool w_blocked = false;
    void finish_w_slabs(bool keep_slabs, const int *done) {
        const T *reg = slabs.p + slab_w_off;
        if (w_blocked) {
            timed("reduce_XHt", 0.0, (double)P * K * (w_nslab + 1) * sizeof(T), [&] {
                hipLaunchKernelGGL(reduce_slabs_blocked_kernel<T>, dim3((unsigned)((P * K + 255) / 256)), dim3(256), 0, stream, numW_p, reg,
                                   P, K, Pc, w_nslab, w_stride, (int64_t)0, P, done);
                HIP_TRY(hipGetLastError());
            });
            w_in_slabs = false;
        } else if (!keep_slabs || w_nslab > 2) {
            reduce_slabs_from("reduce_XHt", numW_p, reg, w_stride, w_nslab, done);
            w_in_slabs = false;
        } else {
            w_in_slabs = true;
        }
    }
    const T *w_num() const { return w_in_slabs ? slabs.p + slab_w_off : numW_p; }
    int w_num_nslab() const {
        if (w_in_slabs && w_nslab > 2) throw StatusError{NMFX_ERR_UNSUPPORTED, "internal: more than two numerator slabs left for the update epilogue"};
        return w_in_slabs ? w_nslab : 1;
    }

    // the two-launch form of [H finalize, W column sums, W finalize, stop rule] for the one-GPU MultUpdate-MSE step (kernels.hpp:
    // col_stats_hfin_kernel, wfin_check_kernel); NMFX_STATS_FUSED=0 (development switch): the four launches
    bool stats_fused_enabled = true;
    DevBuf<double> stat_part_w;
    bool stats_fuse_ok(const nmfx_opts &o) const { return stats_fused_enabled && !sharded() && o.track_objective == 0 && o.stop_sums == 0; }
    void stats_w_check_fused(const T *Wn, const T *Wo, const nmfx_opts &o, long long t, const int *done) {
        stat_part_w.ensure((size_t)stat_chunks_w * 2 * K);
        timed("stats_W_check", 0.0, 2.0 * P * K * sizeof(T), [&] {
            hipLaunchKernelGGL(col_stats_hfin_kernel<T>, dim3(stat_chunks_w, (unsigned)K), dim3(256), 0, stream, Wn, Wo, P, P, (int)K, stat_part_w.p,
                               o.update_H ? stat_part.p : (const double *)nullptr, h_stat_chunks, hstat.p, done);
            // (a thread per output while 2K <= 1024: every partial of the block in flight at once, one round trip)
            hipLaunchKernelGGL(wfin_check_kernel<T>, dim3(1), dim3((unsigned)std::min<int64_t>(1024, std::max<int64_t>(256, 2 * K))), 0, stream, stat_part_w.p, stat_chunks_w, (int)K, wstat.p,
                               o.update_H ? hstat.p : (const double *)nullptr, ctrl, (int)k, (T)o.tol, t, done);
            HIP_TRY(hipGetLastError());
        });
        check_fused = true;
    }
    // (nmfx_opts.stop_sums = 1: the pass of enqueue_check overwrites wstat / hstat with the sequential sums -- the stand-alone tree-sum
    // passes are skipped; partials that come out of an update's epilogue are simply not finalised)
    bool skip_tree_stats = false;
    void stats_w(const T *Wn, const T *Wo, const int *done) {
        if (skip_tree_stats) return;
        timed("stats_W", 0.0, 2.0 * P * K * sizeof(T), [&] {
            hipLaunchKernelGGL(col_stats_kernel<T>, dim3(stat_chunks_w, (unsigned)K), dim3(256), 0, stream, Wn, Wo, P, P,
                               (int)K, stat_part.p, done);
            hipLaunchKernelGGL(finalize_partials_kernel<double>, dim3((unsigned)((2 * K + 3) / 4)), dim3(256), 0, stream,
                               stat_part.p, stat_chunks_w, (int)(2 * K), (int)(2 * K), wstat.p, done);
            HIP_TRY(hipGetLastError());
        });
    }
    // H statistics whose per-r-tile partials were produced by the update GEMM's epilogue (EpiMultUpdate<T,1>)
    void stats_h_finalize(int chunks, const int *done) {
        if (skip_tree_stats) return;
        timed("stats_H", 0.0, (double)chunks * 2 * K * sizeof(double), [&] {
            hipLaunchKernelGGL(finalize_partials_kernel<double>, dim3((unsigned)((2 * K + 3) / 4)), dim3(256), 0, stream,
                               stat_part.p, chunks, (int)(2 * K), (int)(2 * K), hstat.p, done);
            HIP_TRY(hipGetLastError());
        });
    }
    void stats_h(const T *Hn, const T *Ho, const int *done) {
        if (skip_tree_stats) return;
        timed("stats_H", 0.0, 2.0 * K * N * sizeof(T), [&] {
            hipLaunchKernelGGL(row_stats_kernel<T>, dim3(stat_chunks_h), dim3(256), 0, stream, Hn, Ho, N, K, (int)K,
                               stat_part.p, done);
            hipLaunchKernelGGL(finalize_partials_kernel<double>, dim3((unsigned)((2 * K + 3) / 4)), dim3(256), 0, stream,
                               stat_part.p, stat_chunks_h, (int)(2 * K), (int)(2 * K), hstat.p, done);
            HIP_TRY(hipGetLastError());
        });
    }

    // One all-reduce per outer iteration (multi-GPU, replicated W update): numW, gramH and the H statistics.
    void allreduce_w_side(bool with_hstat, const int *done);
    // Row-sharded W side: reduce-scatter of the numerator by row blocks (+ all-reduce of the small k x k / k-vector tail),
    // and the all-gather that re-assembles W (and sums the ranks' column statistics) afterwards.
    void scatter_w_numerator(bool with_hstat, const int *done, bool with_tail = true);
    void gather_w_rows(T *Wfull, bool with_stats, const int *done);
    void stats_w_rows(const T *Wn, const T *Wo, const int *done);

    void enqueue_objective(int alg, const nmfx_opts &o, double *dst, const int *done);
    void enqueue_multmse(const nmfx_opts &o, long long t);
    void multmse_w_rows_fused(const nmfx_opts &o, long long t);
    void multmse_w_rows_fused_peer(const nmfx_opts &o, long long t, PeerComm *pc);
    const CombineDst<T> *peer_dst = nullptr;   // set around times_ht: the peers' receive slots of the numerator (peer-to-peer transport)
    // k <= 64, Float32, one GPU: the 4-launch path of smallk.hpp (smallk_impl.hpp)
    bool smallk_enabled = true;          // NMFX_SMALLK=0 keeps the general path
    bool smallk_attr_set = false;
    bool smallk_grams_valid = false;     // gramW_p / gramH_p hold the Grams of the CURRENT factors (reset by every iterate())
    DevBuf<T> smallk_slabs;              // the stripes' Gram contributions
    DevBuf<unsigned> smallk_ticket;      // arrival counter of the W-side finish launch's statistics blocks (the last one runs the stop rule)
    // measured crossover (scripts/bench: 1024^2 2.4x, 2048^2 1.75x, 4096^2 1.25x faster than the general path; 8192^2 0.8x): a stripe
    // kernel re-reads the whole other factor per 16-wide stripe, which stops paying once the problem is large enough to keep the
    // split-K products busy; very skewed shapes leave one side with too few stripes
    bool smallk_ok() const {
        return sizeof(T) == 4 && K == 64 && !sharded() && smallk_enabled && !use_bf16x3() && P * N <= (int64_t)4096 * 4096 &&
               std::max(P, N) <= 4 * std::min(P, N);
    }
    void enqueue_multmse_smallk(const nmfx_opts &o, long long t);
    void enqueue_multdiv(const nmfx_opts &o, long long t);
    // multdiv on one GPU: the passes behind each numerator product fused (kernels.hpp: div_h_fused_kernel / div_w_fused_kernel);
    // svec / sH_p then carry sum(W, dims=1) / sum(H, dims=2) of the CURRENT factors from one side's pass to the other's
    bool div_fused = true, div_sw_valid = false, div_sh_valid = false;
    bool div_ieee = false;                // NMFX_DIV_IEEE=1: the ratio pass divides with the correctly rounded IEEE sequence (gemm_mfma.hpp: ratio_div_fast)
    void enqueue_projals(const nmfx_opts &o, long long t);
    void spd_factor(T *A, T lambda, T *Uinv, const char *tag_potrf, const char *tag_trtri, const int *done, T *Tm = nullptr);
    // pdsolve!'s potrs! by blocked triangular substitution (chol.hpp: potrs_panel_kernel): the K x NB panel of the right-hand side lives in
    // one workgroup's LDS; larger k keeps the product form Uinv (Uinv' B).  NMFX_POTRS=0 forces the product form (A/B, tests).
    static constexpr int POTRS_NB = sizeof(T) == 4 ? 64 : 32, POTRS_NT = 512;
    bool potrs_enabled = true;            // NMFX_POTRS=0: the product form everywhere (A/B)
    bool potrs_iter = false;              // NMFX_POTRS=1: ProjectedALS's H solve by substitution as well (default: nmfx_pdsolve only)
    // the panel (K x (NB + 1)) and one block column of the packed factor (32 x K) in LDS; 16-byte chunks of a block column: <= 8 per thread
    bool potrs_ok() const {
        return potrs_enabled && K % 64 == 0 && N % POTRS_NB == 0 && (size_t)K * (POTRS_NB + 1 + 32) * sizeof(T) <= (size_t)160 * 1024 &&
               K * 32 / (16 / (int64_t)sizeof(T)) / 512 <= 8 && K * 32 / (16 / (int64_t)sizeof(T)) % 512 == 0;
    }
    // potrs! by strips (chol.hpp: potrs_strip_kernel): one wave per 16 columns, the strip in accumulator registers, the factor packed in the
    // order the sweeps consume it; K / 32 is a compile-time parameter (2, 4, 6, 8; Float32 also 10, 12, 14, 16).  NMFX_POTRS_STRIP=0 (development switch): the panel kernel.
    bool strip_enabled = true;
    // (Float32 up to K = 512: the strip is then 128 registers of a 512-register wave; Float64 strips are twice as wide: K <= 256)
    static constexpr int64_t STRIP_KMAX = sizeof(T) == 4 ? 512 : 256;
    bool strip_ok() const { return strip_enabled && potrs_enabled && K % 64 == 0 && K <= STRIP_KMAX && N % STRIP_COLS == 0; }
    size_t potrs_pack_elems() const { return std::max((size_t)K * K, strip_ok() ? (size_t)strip_pack_elems((int)(K / 32)) : (size_t)0); }
    bool potrs_route_ok() const { return strip_ok() || potrs_ok(); }
    // potrf! with the trailing matrix in registers (chol.hpp: potrf_reg_kernel): K / 32 blocks per side at compile time, Float32 up to 8
    // (k <= 256), Float64 up to 4; beyond that the LDS-panel kernel.  NMFX_POTRF_REG=0 (development switch): the LDS-panel kernel everywhere.
    bool potrf_reg_enabled = true;
    bool potrf_reg_ok() const { return potrf_reg_enabled && K % 64 == 0 && K / 32 <= (sizeof(T) == 4 ? 8 : 4); }
    // the products that share their CUs with the factorisation keep the k-loop unrolled by two when the factorisation is the short
    // register-resident one (launch_gemm_cfg); NMFX_CHOL_UNROLLED=0: the rolled loop as before (A/B)
    bool chol_unrolled = true;
    bool stop_sums_v1 = false;            // NMFX_STOP_SUMS_V1=1 (development switch): the first form of the exact stop sums (16 chains per workgroup, a launch per factor)
    int64_t direct_max_ktiles = 128;      // row-sharded fused step: longest local contraction (in k-tiles) whose X_g H_g' runs unsplit into the PEERS' slots
    bool direct_long_local = true;        // ... no such limit when it goes into the local send buffer (NMFX_DIRECT_LONG=0: the limit there too)
    bool unsplit_enabled = true;          // NMFX_UNSPLIT=0 (development switch): keep the 2-way split of the big products everywhere (solver_impl.hpp: iterate)
    bool xht_images = true;               // NMFX_PROJALS_XT=0 (development switch): ProjectedALS's XH' under the chain on the row-contiguous kernel (A/B)
    bool defer_pack = false;              // set around the H side's factor_under: spd_factor leaves the pack of the factor to the caller
    int spd_solve_left_potrs(const T *Tm, const T *B, T *out, bool clamp, const T *old, const int *done);
    void spd_solve_left(const T *Uinv, const T *B, T *Y, T *out, bool clamp, const int *done);
    void spd_solve_right(const T *Uinv, T *invA, const T *A, T *out, int64_t rows, bool clamp, const int *done);
    // coordinate-descent updaters (cd_impl.hpp)
    void enqueue_cd(const nmfx_opts &o, long long t);
    void enqueue_greedycd(const nmfx_opts &o, long long t);
    template <typename F> void with_kmax(F &&f);
    template <typename F> void with_kmax_greedy(F &&f);
    bool cd_use_lds() const;
    bool cd_force_lds = false;   // NMFX_CD_LDS=1: the LDS forms of the sweeps also for k <= 1024 (tests: bit-identical to the register forms)
    int cd_blocked = -1;                 // NMFX_CD_BLOCKED=0: CoordinateDescent on the row-chain sweep kernels instead of the blocked one (k <= 512)
    void prepare_cd_permutations(const nmfx_opts &o);
    const int *cd_permutation_window(const nmfx_opts &o, long long t);
    static constexpr long long CD_PERM_WINDOW = 256;
    DevBuf<int> greedy_queue;   // GreedyCD's sweep: the row counters of the persistent launch (cd.hpp, GREEDY_NQ x GREEDY_QSTRIDE ints)
    DevBuf<int> cd_perm;   // CoordinateDescent(shuffle = true): the component orders of a window of iterations
    std::vector<int> cd_perm_host;
    long long cd_perm_w0 = -1;
    void cd_sweep_ordered(SampleView<const T> Zo, SampleView<T> Zn, SampleView<const T> Num, const T *Pm, int64_t nsamples, T l1,
                          const int *perm, int64_t offset, const int *done);
    void cd_sweep(SampleView<const T> Zo, SampleView<T> Zn, SampleView<const T> Num, const T *Pm, int64_t nsamples, T l1, const int *done);
    void greedy_side(const char *tag, SampleView<const T> Zo, SampleView<T> Zn, SampleView<const T> G, const T *Pm, int64_t nsamples,
                     T lambda, bool sharded_samples, const int *done);
    void allreduce_hstat(const int *done) {   // CD order: H is updated AFTER the packed W-side all-reduce
        (void)done;
        if (sharded()) comm->all_reduce(hstat.p, (size_t)2 * K, CT_F64, false, stream);
    }
    void enqueue_check(const nmfx_opts &o, long long t) {
        const bool track = o.track_objective != 0;
        if (o.stop_sums != 0) {
            // the reference's sequential T-precision sums (nmfx_opts.stop_sums): the factors of iteration t against those of t - 1 (the
            // ping-pong partners), overwriting the tree sums the update launches left in wstat / hstat
            w_sync(done_flag());
            if (!stop_sums_v1) {   // both factors' chains in one launch, 4 per workgroup (kernels.hpp: stop_sums_exact2_kernel)
                const int nbw = (int)((k + 3) / 4);
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&stop_sums_exact2_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)stop_sums_exact2_lds<T>()));
                hipLaunchKernelGGL((stop_sums_exact2_kernel<T>), dim3((unsigned)(o.update_H ? 2 * nbw : nbw)), dim3(512), stop_sums_exact2_lds<T>(), stream, W[wcur].p, W[wcur ^ 1].p, P,
                                   o.update_H ? H[hcur].p : (const T *)nullptr, o.update_H ? H[hcur ^ 1].p : (const T *)nullptr, N, K, (int)k, nbw, wstat.p,
                                   o.update_H ? hstat.p : (double *)nullptr, done_flag());
            } else {
            hipLaunchKernelGGL((stop_sums_exact_kernel<T, true>), dim3((unsigned)((k + 15) / 16)), dim3(256), 0, stream, W[wcur].p, W[wcur ^ 1].p, P, (int64_t)1, P, (int)k, wstat.p,
                               done_flag());
            if (o.update_H)
                hipLaunchKernelGGL((stop_sums_exact_kernel<T, false>), dim3((unsigned)((k + 15) / 16)), dim3(256), 0, stream, H[hcur].p, H[hcur ^ 1].p, N, K, (int64_t)1, (int)k,
                                   hstat.p, done_flag());
            }
            HIP_TRY(hipGetLastError());
            check_fused = false;
        }
        if (check_fused) {
            check_fused = false;   // stats_check_kernel ran the stop rule of iteration t (never while tracking)
        } else {
            hipLaunchKernelGGL(check_kernel<T>, dim3(1), dim3(256), 0, stream, ctrl, wstat.p,
                               o.update_H ? hstat.p : (const double *)nullptr, (int)k, (T)o.tol, t, track ? dev_trace.p : (double *)nullptr);
            HIP_TRY(hipGetLastError());
        }
        if (track) {   // verbose-style tracking also time-stamps every iteration (common.jl:77: elapsed = time() - start)
            while (iter_events.size() <= (size_t)t) {
                hipEvent_t e;
                HIP_TRY(hipEventCreate(&e));
                iter_events.push_back(e);
            }
            HIP_TRY(hipEventRecord(iter_events[(size_t)t], stream));
        }
    }
    // per-iteration columns of the reference's verbose table for the last tracked solve (common.jl:54-59, :76-82)
    void begin_iter_trace(const nmfx_opts &o) {
        if (!o.track_objective) { iter_trace_len = 0; return; }
        dev_trace.ensure((size_t)o.maxiter + 1);
        std::vector<double> nanv((size_t)o.maxiter + 1, std::nan(""));
        HIP_TRY(hipMemcpyAsync(dev_trace.p, nanv.data(), nanv.size() * sizeof(double), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    void end_iter_trace(const nmfx_opts &o, long long niters) {
        if (!o.track_objective) return;
        iter_trace_len = (int)niters + 1;
        iter_elapsed.assign((size_t)iter_trace_len, 0.0);
        iter_relchange.assign((size_t)iter_trace_len, std::nan(""));
        for (long long t = 1; t <= niters; ++t) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, ev_beg, iter_events[(size_t)t]));
            iter_elapsed[(size_t)t] = ms * 1e-3;
        }
        HIP_TRY(hipMemcpy(iter_relchange.data(), dev_trace.p, (size_t)iter_trace_len * sizeof(double), hipMemcpyDeviceToHost));
    }
    int get_iter_trace(double *elapsed, double *relchange, int count) override {
        const int m = std::min(count, iter_trace_len);
        for (int i = 0; i < m; ++i) {
            if (elapsed) elapsed[i] = iter_elapsed[(size_t)i];
            if (relchange) relchange[i] = iter_relchange[(size_t)i];
        }
        return m;
    }
    const int *done_flag() const { return &ctrl->done; }

    void run_alspgrad(const nmfx_opts &o, nmfx_result *out, double *trace);
    long long w_subsolve(T *Wc, const T *Hc, const nmfx_opts &o, T tolg, long long *inner);
    long long pg_subsolve(bool left, T *Z, const T *Gram, const T *B, int maxiter, int traceiter, T tolg, T beta, T sigma,
                          long long *inner_total);
    struct PgState *pg_state = nullptr, *pg_host = nullptr;
    int pg_refresh_opt = 0;             // nmfx_opts.pg_refresh of the running solve
    static constexpr int PG_REFRESH_F32_DEFAULT = 16;
    int pg_spec_hint[2] = {3, 3};   // speculative line-search steps per enqueued inner iteration (H side, W side), alspgrad_impl.hpp
    DevBuf<double> pg_part;
    long long pg_backtracks = 0;
};

}  // namespace nmfx
