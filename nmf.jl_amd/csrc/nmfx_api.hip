// nmfx_api.hip -- extern "C" entry points of libnmfx.so (see include/nmfx.h).
// Thin: argument checks, dtype dispatch, exception -> status translation.
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>

#include "solver.hpp"

using namespace nmfx;

// Solver<float> / Solver<double> live in translation units of their own (solver_f32.hip, solver_f64.hip)
namespace nmfx {
SolverBase *make_solver_f32(int64_t p, int64_t n_local, int64_t k, int device);
SolverBase *make_solver_f64(int64_t p, int64_t n_local, int64_t k, int device);
}  // namespace nmfx

struct nmfx_ctx {
    SolverBase *impl = nullptr;
    std::string err;
};

static std::string g_create_err;
static std::mutex g_create_mu;

template <typename F> static int guarded(nmfx_ctx *ctx, F &&f) {
    std::string *err = ctx ? &ctx->err : &g_create_err;
    try {
        f();
        err->clear();
        return NMFX_OK;
    } catch (const StatusError &e) {
        *err = e.msg;
        return e.status;
    } catch (const HipError &e) {
        *err = std::string("HIP error: ") + hipGetErrorString(e.e) + " at " + e.what + " (solver line " + std::to_string(e.line) + ")";
        return NMFX_ERR_HIP;
    } catch (const RcclError &e) {
        *err = std::string("RCCL error: ") + ncclGetErrorString(e.e) + " at " + e.what;
        return NMFX_ERR_RCCL;
    } catch (const RcclFailure &e) {
        *err = std::string("RCCL error: ") + ncclGetErrorString(e.e) + " at " + e.what;
        return NMFX_ERR_RCCL;
    } catch (const CommError &e) {
        *err = std::string("communicator error: ") + e.msg;
        return NMFX_ERR_RCCL;
    } catch (const std::exception &e) {
        *err = e.what();
        return NMFX_ERR_STATE;
    }
}

extern "C" {

const char *nmfx_version(void) { return "nmfx 0.2 (gfx950, hand-written HIP/MFMA)"; }

const char *nmfx_last_error(const nmfx_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int nmfx_create(nmfx_ctx **out, int dtype, int64_t p, int64_t n_local, int64_t k, int device) {
    std::lock_guard<std::mutex> lk(g_create_mu);
    if (!out) { g_create_err = "out is NULL"; return NMFX_ERR_BAD_ARG; }
    *out = nullptr;
    if (dtype != NMFX_F32 && dtype != NMFX_F64) { g_create_err = "dtype must be NMFX_F32 or NMFX_F64"; return NMFX_ERR_BAD_ARG; }
    if (p < 1 || n_local < 1 || k < 1) { g_create_err = "p, n, k must be positive"; return NMFX_ERR_DIM_MISMATCH; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        g_create_err = "no HIP device visible: libnmfx has no CPU fallback";
        return NMFX_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) { g_create_err = "device ordinal out of range"; return NMFX_ERR_BAD_ARG; }
    nmfx_ctx *c = new nmfx_ctx;
    int st = guarded(nullptr, [&] {
        if (dtype == NMFX_F32) c->impl = make_solver_f32(p, n_local, k, device);
        else c->impl = make_solver_f64(p, n_local, k, device);
    });
    if (st != NMFX_OK) { delete c; return st; }
    *out = c;
    return NMFX_OK;
}

void nmfx_destroy(nmfx_ctx *ctx) {
    if (!ctx) return;
    delete ctx->impl;
    delete ctx;
}

int nmfx_set_X(nmfx_ctx *ctx, const void *X_host, int64_t ldx) {
    if (!ctx || !X_host) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] { ctx->impl->set_X(X_host, ldx, false); });
}

int nmfx_set_X_device(nmfx_ctx *ctx, const void *X_dev, int64_t ldx) {
    if (!ctx || !X_dev) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] { ctx->impl->set_X(X_dev, ldx, true); });
}

int nmfx_set_factors(nmfx_ctx *ctx, const void *W_host, const void *H_host) {
    if (!ctx || !W_host || !H_host) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] { ctx->impl->set_factors(W_host, H_host); });
}

int nmfx_get_factors(nmfx_ctx *ctx, void *W_host, void *H_host) {
    if (!ctx) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] { ctx->impl->get_factors(W_host, H_host); });
}

int nmfx_iterate(nmfx_ctx *ctx, int alg, const nmfx_opts *opts, nmfx_result *out, double *objv_trace) {
    if (!ctx || !opts || !out) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] { ctx->impl->iterate(alg, *opts, out, objv_trace); });
}

int nmfx_solve(nmfx_ctx *ctx, int alg, const nmfx_opts *opts, void *W_host, void *H_host, nmfx_result *out,
               double *objv_trace) {
    if (!ctx || !opts || !out || !W_host || !H_host) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        ctx->impl->set_factors(W_host, H_host);
        ctx->impl->iterate(alg, *opts, out, objv_trace);
        // update_H == 0: H must come back bit-identical (test/interf.jl:33-37) -> do not even copy it
        ctx->impl->get_factors(W_host, opts->update_H ? H_host : nullptr);
    });
}

int nmfx_alspgrad_subsolve(nmfx_ctx *ctx, int which, const nmfx_opts *opts, void *W_host, void *H_host,
                           nmfx_result *out) {
    if (!ctx || !opts || !out || !W_host || !H_host || (which != 0 && which != 1)) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        ctx->impl->set_factors(W_host, H_host);
        ctx->impl->subsolve(which, *opts, out);
        ctx->impl->get_factors(which == 1 ? W_host : nullptr, which == 0 ? H_host : nullptr);
    });
}

int nmfx_comm_get_unique_id(void *out_bytes) {
    if (!out_bytes) return NMFX_ERR_BAD_ARG;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return NMFX_ERR_RCCL;
    std::memset(out_bytes, 0, NMFX_UNIQUE_ID_BYTES);
    std::memcpy(out_bytes, &id, sizeof id);
    return NMFX_OK;
}

int nmfx_comm_init(nmfx_ctx *ctx, const void *unique_id_bytes, int rank, int nranks) {
    if (!ctx || !unique_id_bytes || nranks < 1 || rank < 0 || rank >= nranks) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] { ctx->impl->comm_init(unique_id_bytes, rank, nranks); });
}

int nmfx_local_group_create(nmfx_local_group **out, int nranks) {
    if (!out || nranks < 1 || nranks > LOCAL_MAX_RANKS) return NMFX_ERR_BAD_ARG;
    *out = reinterpret_cast<nmfx_local_group *>(new LocalGroup(nranks));
    return NMFX_OK;
}

void nmfx_local_group_destroy(nmfx_local_group *group) {
    if (group) LocalGroup::release(reinterpret_cast<LocalGroup *>(group));   // deferred while contexts are still attached
}

int nmfx_comm_init_local(nmfx_ctx *ctx, nmfx_local_group *group, int rank) {
    LocalGroup *g = reinterpret_cast<LocalGroup *>(group);
    if (!ctx || !g || rank < 0 || rank >= g->n) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] { ctx->impl->comm_init_local(g, rank); });
}

int nmfx_comm_init_sim(nmfx_ctx *ctx, int rank, int nranks) {
    if (!ctx || nranks < 1 || rank < 0 || rank >= nranks) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] { ctx->impl->comm_init_sim(rank, nranks); });
}

int nmfx_comm_init_p2p(nmfx_ctx *ctx, int rank, int nranks) {
    if (!ctx) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] { ctx->impl->comm_init_p2p(rank, nranks); });
}

int nmfx_comm_p2p_export(nmfx_ctx *ctx, void *handle_out) {
    if (!ctx || !handle_out) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] { ctx->impl->p2p_export(handle_out); });
}

int nmfx_comm_p2p_attach(nmfx_ctx *ctx, const void *all_handles) {
    if (!ctx) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] { ctx->impl->p2p_attach(all_handles); });   // all_handles = NULL: detach (back to the wrapped transport)
}

int nmfx_comm_p2p_stats(nmfx_ctx *ctx, int64_t *served_by_windows, int64_t *served_by_base) {
    if (!ctx) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        long long w = 0, b = 0;
        ctx->impl->p2p_stats(&w, &b);
        if (served_by_windows) *served_by_windows = w;
        if (served_by_base) *served_by_base = b;
    });
}

int nmfx_comm_set_mode(nmfx_ctx *ctx, int mode) {
    if (!ctx || mode < NMFX_COMM_ROW_SHARDED || mode > NMFX_COMM_REPLICAS) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] { ctx->impl->comm_set_mode(mode); });
}

int nmfx_pdsolve(nmfx_ctx *ctx, const void *A_host, double lambda, const void *B_host, void *X_host, int project_nn) {
    if (!ctx || !A_host || !B_host || !X_host) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] { ctx->impl->pdsolve_host(0, A_host, B_host, lambda, X_host, project_nn != 0); });
}

int nmfx_pdrsolve(nmfx_ctx *ctx, const void *A_host, const void *B_host, double lambda, void *X_host, int project_nn) {
    if (!ctx || !A_host || !B_host || !X_host) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] { ctx->impl->pdsolve_host(1, A_host, B_host, lambda, X_host, project_nn != 0); });
}

int nmfx_spa_init(nmfx_ctx *ctx, int warm_sweeps, int64_t *anchors_out, int64_t *unsolved_out) {
    if (!ctx) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] { ctx->impl->spa_init(warm_sweeps, anchors_out, unsolved_out); });
}

int nmfx_objective(nmfx_ctx *ctx, int alg, const nmfx_opts *opts, double *out) {
    if (!ctx || !opts || !out) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] { *out = ctx->impl->objective(alg, *opts); });
}

int nmfx_check_nonneg(nmfx_ctx *ctx, int which, int *all_nonneg) {
    if (!ctx || !all_nonneg) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] { *all_nonneg = ctx->impl->check_nonneg(which) ? 1 : 0; });
}

int nmfx_randinit(nmfx_ctx *ctx, uint64_t seed, int normalize, int zeroh, int64_t h_col_offset) {
    if (!ctx || h_col_offset < 0) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] { ctx->impl->randinit(seed, normalize != 0, zeroh != 0, h_col_offset); });
}

int nmfx_solve_replicates(nmfx_ctx *ctx, int alg, const nmfx_opts *opts, int replicates, uint64_t seed, int zeroh,
                          int64_t h_col_offset, void *W_host, void *H_host, nmfx_result *out, int *best_replicate) {
    if (!ctx || !opts || !out || !W_host || !H_host || h_col_offset < 0) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        ctx->impl->solve_replicates(alg, *opts, replicates, seed, zeroh != 0, h_col_offset, W_host, H_host, out, best_replicate);
    });
}

int nmfx_nndsvd(nmfx_ctx *ctx, const void *U_host, const void *s_host, const void *V_host, int variant, int zeroh, uint64_t seed,
                int64_t n_total) {
    if (!ctx) return NMFX_ERR_BAD_ARG;
    if ((U_host == nullptr) != (s_host == nullptr) || (U_host == nullptr) != (V_host == nullptr)) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] { ctx->impl->nndsvd_init(U_host, s_host, V_host, variant, zeroh != 0, seed, n_total); });
}

int nmfx_rsvd_begin(nmfx_ctx *ctx, uint64_t seed, int64_t h_col_offset, int power_iters, void *C_host) {
    if (!ctx || !C_host || h_col_offset < 0) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] { ctx->impl->rsvd_begin(seed, h_col_offset, power_iters, C_host); });
}

int nmfx_rsvd_finish(nmfx_ctx *ctx, const void *Ub_host, const void *s_host, void *U_out, void *Vt_out) {
    if (!ctx || !Ub_host || !s_host) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] { ctx->impl->rsvd_finish(Ub_host, s_host, U_out, Vt_out); });
}

int nmfx_get_iter_trace(nmfx_ctx *ctx, double *elapsed_s, double *relchange, int count, int *n_entries) {
    if (!ctx || !n_entries || count < 0) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] { *n_entries = ctx->impl->get_iter_trace(elapsed_s, relchange, count); });
}

int nmfx_profile_enable(nmfx_ctx *ctx, int on) {
    if (!ctx) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] { ctx->impl->profile_enable(on); });
}

int nmfx_set_final_objective(nmfx_ctx *ctx, int on) {
    if (!ctx) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] { ctx->impl->set_final_objective(on != 0); });
}

int nmfx_profile_get(nmfx_ctx *ctx, nmfx_kernel_stat *out, int max_entries, int *n_entries) {
    if (!ctx || !out || !n_entries) return NMFX_ERR_BAD_ARG;
    return guarded(ctx, [&] { *n_entries = ctx->impl->profile_get(out, max_entries); });
}

int nmfx_device_info(int device, char *name_out, int name_len, int *cu_count, int64_t *hbm_bytes) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return NMFX_ERR_NO_DEVICE;
    if (name_out && name_len > 0) std::snprintf(name_out, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return NMFX_OK;
}

}  // extern "C"

// explicit instantiation
