// alspgrad_impl.hpp -- ALSPGrad (src/alspgrad.jl:352-425): alternating projected-gradient sub-solves.
// The outer loop and the inner-iteration count are host-driven (the inner iteration count feeds the
// `tolg *= 0.1` rule, :409-421); back-tracking steps are batched on the device (pgrad.hpp).
#pragma once
#include "pgrad.hpp"
#include "solver.hpp"

namespace nmfx {

// returns executed inner iterations; Z updated in place
template <typename T>
long long Solver<T>::pg_subsolve(bool left, T *Z, const T *Gram, const T *B, int maxiter, int traceiter, T tolg, T beta,
                                 T sigma, long long *inner_total) {
    // left: Z = H (K x N, this rank's columns).  right: Z = W -- all P rows, or, when the W side is row-sharded, the
    // rank's row block [row0, row0 + Pc): the caller passes Z and B already offset by row0 (ld stays P), every rank solves
    // its rows with the SAME global step size (the scalars of the line search are all-reduced, exactly like on the H side).
    const bool wrows = !left && row_sharded();
    const int64_t Rw = wrows ? Pc : P;                                     // rows of the W block this call works on
    const int64_t rows = left ? K : Rw, cols = left ? N : K, ldz = left ? K : P;
    // the rotating (Z, G) sets (pgrad.hpp): three of each, every set with the layout and leading dimension of Z itself
    const int64_t set_stride = std::max<int64_t>((int64_t)K * N, (int64_t)P * K);
    work[3].ensure((size_t)3 * set_stride);
    work[4].ensure((size_t)3 * set_stride);
    T *GS = work[3].p + (wrows ? row0 : 0);
    T *ZS = work[4].p + (wrows ? row0 : 0);
    pg_part.ensure((size_t)PG_NSUM * 65536 + 8);
    double *pg_red = pg_part.p + (size_t)PG_NSUM * 65536;                  // the rank-local sums on their way through a collective
    if (!pg_state) {
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&pg_state), sizeof(PgState)));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&pg_host), sizeof(PgState)));
    }
    PgState init;
    std::memset(&init, 0, sizeof init);
    init.alpha = 1.0;                                                      // :119 alpha = 1 at entry
    init.idle = 1;
    HIP_TRY(hipMemcpyAsync(pg_state, &init, sizeof init, hipMemcpyHostToDevice, stream));
    const bool reduce_scalars = (left && sharded()) || wrows;                    // H is column-sharded; W row-sharded or replicated
    const bool tiny_ok = reduce_scalars && comm->tiny_capable();                 // peer transport: the scalars travel inside the decision kernels
    const T epsT = std::numeric_limits<T>::epsilon();
    const int *idle = &pg_state->idle, *gate = &pg_state->gate;
    // ~2048 blocks: row chunks of >= 1024 rows, the rest of the parallelism from the columns
    const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(64, rows / 1024));
    const unsigned gy = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cols, 2048 / gx));
    // Z -> set 0 (base = 0 in the fresh state)
    hipLaunchKernelGGL(pg_copy_kernel<T>, dim3(gx, gy), dim3(256), 0, stream, Z, ZS, set_stride, rows, cols, ldz, pg_state, 0);
    HIP_TRY(hipGetLastError());
    // G = Gram*Z - B  (+ projgradnorm^2 partials) of the CURRENT set            :124-130 / :280-286
    auto grad = [&]() {
        EpiGradNorm<T> e{B, ZS, GS, left ? K : P, pg_part.p, 0.0};
        e.st = pg_state; e.set_stride = set_stride;
        Seg sg;
        sg.sel = &pg_state->base;
        if (left) { sg.a_sel = set_stride; gemm<KCONTIG, KCONTIG>("gemm_pg_grad", ZS, K, N, Gram, K, K, K, 1, true, e, gate, 4.0 * K * N * sizeof(T), sg); }
        else { sg.b_sel = set_stride; gemm<KSTRIDED, KSTRIDED>("gemm_pg_grad", Gram, K, K, ZS, P, Rw, K, 1, false, e, gate, 4.0 * Rw * K * sizeof(T), sg); }
        return last_blocks;
    };
    // one back-tracking step: Gram * D(alpha) with D formed in the operand loader from the current set, the trial point and its gradient
    // written by the epilogue into one of the other two sets, the four scalars reduced there as well
    auto step = [&](bool last_enqueued) {
        EpiPgStep<T> e{ZS, GS, set_stride, left ? K : P, pg_state, pg_part.p, (T)0, (T)0, 0, 0.0, 0.0, 0.0, 0.0};
        Seg sg;
        sg.alpha_ptr = &pg_state->alpha;
        sg.sel = &pg_state->base;
        int nblk;
        if (left) {
            sg.a_aux = GS; sg.a_sel = set_stride; sg.aaux_sel = set_stride;
            gemm<KCONTIG, KCONTIG, 1>("gemm_pg_step", ZS, K, N, Gram, K, K, K, 1, true, e, idle, 5.0 * K * N * sizeof(T), sg);
        } else {
            sg.b_aux = GS; sg.b_sel = set_stride; sg.baux_sel = set_stride;
            gemm<KSTRIDED, KSTRIDED, 2>("gemm_pg_step", Gram, K, K, ZS, P, Rw, K, 1, false, e, idle, 5.0 * Rw * K * sizeof(T), sg);
        }
        nblk = last_blocks;
        // sharded: the four sums are global (ONE step size for all of Z, alspgrad.jl:150-155).  Peer transport: the decision kernel
        // exchanges the rank-local sums itself (16-byte tagged granules, no collective launch).  Other transports: local sums by a
        // one-block launch, all-reduce of 4 doubles (a count that does not depend on the rank's block grid), then the decision.
        if (reduce_scalars && !tiny_ok) {
            hipLaunchKernelGGL(pg_local_sum_kernel, dim3(1), dim3(256), 0, stream, pg_part.p, nblk, PG_NSUM, pg_red);
            comm->all_reduce(pg_red, PG_NSUM, CT_F64, false, stream);
            hipLaunchKernelGGL(pg_decide_kernel<T>, dim3(1), dim3(256), 0, stream, pg_state, pg_red, 1, beta, sigma, epsT, traceiter, last_enqueued ? 1 : 0, TinyAR());
        } else {
            hipLaunchKernelGGL(pg_decide_kernel<T>, dim3(1), dim3(256), 0, stream, pg_state, pg_part.p, nblk, beta, sigma, epsT, traceiter, last_enqueued ? 1 : 0,
                               reduce_scalars ? comm->tiny() : TinyAR());
        }
    };
    // A full product G = Gram*Z - B every REFRESH inner iterations bounds the rounding drift of the running sum
    // (the first iteration of a sub-solve always: Gram and B are new)
    // nmfx_opts.pg_refresh: 1 = the reference's form (a full product at the top of every inner iteration, src/alspgrad.jl:124-127,
    // 280-283); 0 = the library default
    const int REFRESH = pg_refresh_opt > 0 ? pg_refresh_opt : ((sizeof(T) == 4) ? PG_REFRESH_F32_DEFAULT : 64);
    auto fetch = [&]() {
        HIP_TRY(hipMemcpyAsync(pg_host, pg_state, sizeof(PgState), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    };
    // The sub-solve is enqueued AHEAD of the host's knowledge, AHEAD inner iterations at a time, each with SPEC back-tracking
    // steps (the typical search takes 2-3): the state machine on the device turns everything behind a converged iteration
    // into no-ops, and an iteration whose search needs more than SPEC steps raises `halt`, which freezes the iterations
    // enqueued behind it until the host has finished that search step by step.  One host round trip per AHEAD inner iterations
    // instead of one per iteration (it cost ~15 % of the C5 shard's time); the executed sequence, hence every counter and
    // every bit of Z, is the same.
    // SPEC follows the searches seen so far (PgState::hist): the smallest count that all but ~3 % of them fit in -- a launch
    // behind a finished search is a ~4.5 us no-op (x 2 with its decision kernel), a halt costs the rest of the batch as no-ops plus
    // a host round trip (~170 us).  At the C5 shard shape every search takes 2 or 3 steps except 1-2 per sub-solve that take 5+:
    // 3 instead of a fixed 4 saves 400 no-op pairs per outer iteration.  Scheduling only: the executed sequence does not change.
    const int AHEAD = 8;
    int SPEC = std::min(pg_spec_hint[left ? 0 : 1], traceiter);
    int forced = 0;                                                        // NMFX_PG_SPEC=n pins it (tests: every value gives the same bits)
    if (const char *e = dev_env("NMFX_PG_SPEC")) forced = std::atoi(e);
    if (forced > 0) SPEC = std::min(forced, traceiter);
    auto retune = [&]() {
        if (forced > 0) return;
        long long total = 0;
        for (int i = 0; i < 8; ++i) total += pg_host->hist[i];
        if (total < 8) return;
        for (int sp = 2; sp <= 4; ++sp) {
            long long longer = 0;
            for (int i = sp; i < 8; ++i) longer += pg_host->hist[i];
            if (longer * 32 <= total || sp == 4) { SPEC = std::min(sp, traceiter); break; }
        }
        pg_spec_hint[left ? 0 : 1] = SPEC;
    };
    long long t = 0;
    bool converged = false;
    while (!converged && t < maxiter) {
        const int batch = (int)std::min<long long>(AHEAD, (long long)maxiter - t);
        for (int b = 0; b < batch; ++b) {
            // (the accepted trial point of the search before already IS the current set, its projgradnorm^2 sits in PgState::red[3]:
            // between two full products an inner iteration starts without any pass over Z and G)
            const int nblk = ((t + b) % REFRESH == 0) ? grad() : 0;
            if (nblk == 0) {
                hipLaunchKernelGGL(pg_begin_kernel<T>, dim3(1), dim3(256), 0, stream, pg_state, pg_part.p, 0, tolg, TinyAR());
            } else if (reduce_scalars && !tiny_ok) {
                hipLaunchKernelGGL(pg_local_sum_kernel, dim3(1), dim3(256), 0, stream, pg_part.p, nblk, 1, pg_red);
                comm->all_reduce(pg_red, 1, CT_F64, false, stream);
                hipLaunchKernelGGL(pg_begin_kernel<T>, dim3(1), dim3(256), 0, stream, pg_state, pg_red, 1, tolg, TinyAR());
            } else {
                hipLaunchKernelGGL(pg_begin_kernel<T>, dim3(1), dim3(256), 0, stream, pg_state, pg_part.p, nblk, tolg, reduce_scalars ? comm->tiny() : TinyAR());
            }
            for (int sidx = 0; sidx < SPEC; ++sidx) step(sidx == SPEC - 1);
        }
        fetch();
        while (pg_host->halt && !pg_host->nonfinite) {
            // finish the halted search: the remaining steps, a few at a time
            while (!pg_host->idle && pg_host->it < traceiter) {
                const int more = std::min(4, traceiter - pg_host->it);
                for (int sidx = 0; sidx < more; ++sidx) step(false);
                fetch();
            }
            hipLaunchKernelGGL(pg_resume_kernel, dim3(1), dim3(1), 0, stream, pg_state);
            // the iterations that were frozen behind the halt have to be enqueued again: leave the batch loop
            break;
        }
        if (pg_host->nonfinite) throw StatusError{NMFX_ERR_ALPHA_NONFINITE, "alpha is not finite"};
        t = pg_host->t_inner;
        converged = pg_host->converged != 0;
        retune();
    }
    // the current set back into Z (the accept of the last executed search, if any, is part of it)
    hipLaunchKernelGGL(pg_copy_kernel<T>, dim3(gx, gy), dim3(256), 0, stream, Z, ZS, set_stride, rows, cols, ldz, pg_state, 1);
    HIP_TRY(hipGetLastError());
    if (const char *e = dev_env("NMFX_PG_HIST"); e && e[0] == '1' && pg_host) {
        std::fprintf(stderr, "[nmfx] pg_subsolve(%s): %lld inner iterations, searches by steps:", left ? "H" : "W", t);
        for (int i = 0; i < 8; ++i) std::fprintf(stderr, " %d", pg_host->hist[i]);
        std::fprintf(stderr, "\n");
    }
    if (inner_total) *inner_total += t;
    pg_backtracks += pg_host ? pg_host->backtracks : 0;
    return t;
}

// set_h! (src/alspgrad.jl:218-222) + _alspgrad_updatew! (:242-347).  Multi-GPU: the numerator X_g H_g' and H_g H_g' are
// sums over the column shards; the sub-problem itself is row-separable (G = W*HHt - XHt row by row, one GLOBAL step size),
// so with the row-sharded W side every rank solves its Pc rows (2 Pc k^2 per product instead of 2 p k^2 on every rank --
// SURVEY.md section 8e "alspgrad specifics") and the rows are all-gathered afterwards.
template <typename T> long long Solver<T>::w_subsolve(T *Wc, const T *Hc, const nmfx_opts &o, T tolg, long long *inner) {
    const bool rs = row_sharded();
    w_blocked = rs;
    times_ht(X.p, Hc, true, nullptr);
    w_blocked = false;
    if (!rs) {
        allreduce_w_side(false, nullptr);
        return pg_subsolve(false, Wc, gramH_p, numW_p, o.maxsubiter, o.traceiter, tolg, (T)o.beta, (T)o.sigma, inner);
    }
    scatter_w_numerator(false, nullptr);
    const long long it = pg_subsolve(false, Wc + row0, gramH_p, numW_p + row0, o.maxsubiter, o.traceiter, tolg, (T)o.beta, (T)o.sigma, inner);
    gather_w_rows(Wc, false, nullptr);
    return it;
}

template <typename T> void Solver<T>::subsolve(int which, const nmfx_opts &o, nmfx_result *out) {
    precision = o.precision;
    pg_refresh_opt = o.pg_refresh;
    rsvd_ready = 0;
    require_ready();
    HIP_TRY(hipSetDevice(device));
    std::memset(out, 0, sizeof *out);
    pg_backtracks = 0;
    long long inner = 0;
    if (which == 0) {                                                      // alspgrad_updateh! (:69-84)
        wt_times(W[wcur].p, X.p, true, nullptr);                           // set_w! (:63-67)
        out->niters = pg_subsolve(true, H[hcur].p, gramW_p, numH_p, o.maxsubiter, o.traceiter, (T)o.tolg, (T)o.beta, (T)o.sigma, &inner);
    } else {                                                               // alspgrad_updatew! (:225-240)
        out->niters = w_subsolve(W[wcur].p, H[hcur].p, o, (T)o.tolg, &inner);
    }
    HIP_TRY(hipStreamSynchronize(stream));
    if (comm) comm->health();
    out->inner_iters = inner;
    out->backtracks = pg_backtracks;
}

// update_wh!(::ALSPGradUpd) (:400-425) inside nmf_skeleton! (src/common.jl:45-89)
template <typename T> void Solver<T>::run_alspgrad(const nmfx_opts &o, nmfx_result *out, double *trace) {
    const bool track = o.track_objective != 0;
    Ctrl init;
    std::memset(&init, 0, sizeof init);
    HIP_TRY(hipMemcpyAsync(ctrl, &init, sizeof init, hipMemcpyHostToDevice, stream));
    if (track) {
        trace_dev.ensure((size_t)o.maxiter + 1);
        std::vector<double> nanv((size_t)o.maxiter + 1, std::nan(""));
        HIP_TRY(hipMemcpyAsync(trace_dev.p, nanv.data(), nanv.size() * sizeof(double), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    pg_backtracks = 0;
    pg_refresh_opt = o.pg_refresh;
    long long inner = 0;
    T tolg = (T)o.tolg;                                                    // fresh ALSPGradUpd per solve! (:381-383)
    begin_iter_trace(o);
    HIP_TRY(hipEventRecord(ev_beg, stream));
    if (track) enqueue_objective(NMFX_ALG_ALSPGRAD, o, trace_dev.p, nullptr);
    long long t = 0;
    bool converged = false;
    while (!converged && t < o.maxiter) {
        ++t;
        T *Wc = W[wcur].p, *Hc = H[hcur].p;
        T *preW = W[wcur ^ 1].p, *preH = H[hcur ^ 1].p;                    // copyto!(preW, W); copyto!(preH, H)  (common.jl:66-67)
        HIP_TRY(hipMemcpyAsync(preW, Wc, W[0].count * sizeof(T), hipMemcpyDeviceToDevice, stream));
        if (o.update_H) {
            HIP_TRY(hipMemcpyAsync(preH, Hc, H[0].count * sizeof(T), hipMemcpyDeviceToDevice, stream));
            wt_times(Wc, X.p, true, nullptr);                              // set_w! (:405)
            const long long itH = pg_subsolve(true, Hc, gramW_p, numH_p, o.maxsubiter, o.traceiter, tolg, (T)o.beta, (T)o.sigma, &inner);
            if (itH == 1) tolg = (T)((double)tolg * 0.1);                  // :409-411
        }
        const long long itW = w_subsolve(Wc, Hc, o, tolg, &inner);         // set_h! (:415) + _alspgrad_updatew! (:416-417)
        if (itW == 1) tolg = (T)((double)tolg * 0.1);                      // :419-421
        if (o.update_H) {
            stats_h(Hc, preH, nullptr);
            if (sharded()) comm->all_reduce(hstat.p, (size_t)2 * K, CT_F64, false, stream);
        }
        stats_w(Wc, preW, nullptr);
        enqueue_check(o, t);
        if (track) enqueue_objective(NMFX_ALG_ALSPGRAD, o, trace_dev.p + t, nullptr);
        HIP_TRY(hipMemcpyAsync(ctrl_host, ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        converged = ctrl_host->converged != 0;
    }
    if (!track) enqueue_objective(NMFX_ALG_ALSPGRAD, o, obj_final.p, nullptr);
    HIP_TRY(hipEventRecord(ev_end, stream));
    double objv = std::nan("");
    if (track) {
        HIP_TRY(hipMemcpyAsync(&objv, trace_dev.p + t, sizeof(double), hipMemcpyDeviceToHost, stream));
        if (trace) HIP_TRY(hipMemcpyAsync(trace, trace_dev.p, ((size_t)o.maxiter + 1) * sizeof(double), hipMemcpyDeviceToHost, stream));
    } else {
        HIP_TRY(hipMemcpyAsync(&objv, obj_final.p, sizeof(double), hipMemcpyDeviceToHost, stream));
    }
    HIP_TRY(hipStreamSynchronize(stream));
    if (comm) comm->health();
    end_iter_trace(o, t);
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ev_beg, ev_end));
    out->niters = t;
    out->converged = converged ? 1 : 0;
    out->status = 0;
    out->objvalue = objv;
    out->seconds_loop = ms * 1e-3;
    out->inner_iters = inner;
    out->backtracks = pg_backtracks;
    out->final_tolg = (double)tolg;
}

}  // namespace nmfx
