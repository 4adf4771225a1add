// alspgrad_impl.hpp -- ALSPGrad (src/alspgrad.jl:352-425): alternating projected-gradient sub-solves.
// The outer loop and the inner-iteration count are host-driven (the inner iteration count feeds the
// `tolg *= 0.1` rule, :409-421); back-tracking steps are batched on the device (pgrad.hpp).
#pragma once
#include "pgrad.hpp"
#include "solver.hpp"

namespace nmfx {

template <typename T> struct PgBuffers {
    T *G, *Zn, *Zp, *D, *GD;
};

// returns executed inner iterations; Z updated in place
template <typename T>
long long Solver<T>::pg_subsolve(bool left, T *Z, const T *Gram, const T *B, int maxiter, int traceiter, T tolg, T beta,
                                 T sigma, long long *inner_total) {
    const int64_t rows = left ? K : P, cols = left ? N : K;
    const int64_t count = rows * cols;
    const size_t need = (size_t)std::max<int64_t>((int64_t)K * N, (int64_t)P * K);
    for (int i = 3; i < 8; ++i) work[i].ensure(need);
    T *G = work[3].p, *Zn = work[4].p, *Zp = work[5].p, *D = work[6].p, *GD = work[7].p;
    const int NB = 512;
    pg_part.ensure((size_t)2 * NB);
    if (!pg_state) {
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&pg_state), sizeof(PgState)));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&pg_host), sizeof(PgState)));
    }
    PgState init;
    std::memset(&init, 0, sizeof init);
    init.alpha = 1.0;                                                      // :119 alpha = 1 at entry
    init.idle = 1;
    HIP_TRY(hipMemcpyAsync(pg_state, &init, sizeof init, hipMemcpyHostToDevice, stream));
    const bool sharded = left && nranks > 1;                               // H is column-sharded, W is replicated
    const T epsT = std::numeric_limits<T>::epsilon();
    int step_id = 0;
    auto grad = [&](const T *src, T *dst, const T *sub, const int *idle) {
        if (left) {
            if (sub) { EpiSubStore<T> e{sub, dst, K}; gemm<KCONTIG, KCONTIG>("gemm_pg_GramZ", src, K, N, Gram, K, K, K, 1, true, e, idle); }
            else { EpiStore<T> e{dst, K, 0, nullptr}; gemm<KCONTIG, KCONTIG>("gemm_pg_GramD", src, K, N, Gram, K, K, K, 1, true, e, idle); }
        } else {
            if (sub) { EpiSubStore<T> e{sub, dst, P}; gemm<KSTRIDED, KSTRIDED>("gemm_pg_ZGram", Gram, K, K, src, P, P, K, 1, false, e, idle); }
            else { EpiStore<T> e{dst, P, 0, nullptr}; gemm<KSTRIDED, KSTRIDED>("gemm_pg_DGram", Gram, K, K, src, P, P, K, 1, false, e, idle); }
        }
    };
    auto enqueue_steps = [&](int nsteps) {
        const int *idle = &pg_state->idle;
        for (int s = 0; s < nsteps; ++s) {
            ++step_id;
            hipLaunchKernelGGL(pg_project_kernel<T>, dim3(NB), dim3(256), 0, stream, Z, G, Zn, D, count, pg_state, pg_part.p);
            hipLaunchKernelGGL(pg_sum_kernel, dim3(1), dim3(256), 0, stream, pg_part.p, NB, 1, pg_state->red, 0, idle);
            grad(D, GD, nullptr, idle);                                    // :151 WtWD = WtW * D
            hipLaunchKernelGGL(pg_dots_kernel<T>, dim3(NB), dim3(256), 0, stream, GD, D, Z, Zp, Zn, count, pg_state, pg_part.p);
            hipLaunchKernelGGL(pg_sum_kernel, dim3(1), dim3(256), 0, stream, pg_part.p, NB, 2, pg_state->red, 1, idle);
            if (sharded) RCCL_TRY(ncclAllReduce(pg_state->red, pg_state->red, 3, ncclDouble, ncclSum, comm, stream));
            hipLaunchKernelGGL(pg_decide_kernel<T>, dim3(1), dim3(1), 0, stream, pg_state, beta, sigma, epsT, traceiter, step_id);
            hipLaunchKernelGGL(pg_apply_kernel<T>, dim3(NB), dim3(256), 0, stream, Z, Zp, Zn, count, pg_state, step_id);
        }
        HIP_TRY(hipGetLastError());
    };
    auto fetch = [&]() {
        HIP_TRY(hipMemcpyAsync(pg_host, pg_state, sizeof(PgState), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    };
    long long t = 0;
    bool converged = false;
    while (!converged && t < maxiter) {
        ++t;
        grad(Z, G, B, nullptr);                                            // :124-127 G = WtW*H - WtX
        hipLaunchKernelGGL(pg_norm_kernel<T>, dim3(NB), dim3(256), 0, stream, G, Z, count, pg_part.p);
        hipLaunchKernelGGL(pg_sum_kernel, dim3(1), dim3(256), 0, stream, pg_part.p, NB, 1, pg_state->red, 3, (const int *)nullptr);
        if (sharded) RCCL_TRY(ncclAllReduce(pg_state->red + 3, pg_state->red + 3, 1, ncclDouble, ncclSum, comm, stream));
        hipLaunchKernelGGL(pg_begin_kernel<T>, dim3(1), dim3(1), 0, stream, pg_state, tolg);
        enqueue_steps(std::min(3, traceiter));
        fetch();
        int enq = std::min(3, traceiter);
        while (!pg_host->idle && enq < traceiter) {
            const int more = std::min(4, traceiter - enq);
            enqueue_steps(more);
            enq += more;
            fetch();
        }
        if (pg_host->nonfinite) throw StatusError{NMFX_ERR_ALPHA_NONFINITE, "alpha is not finite"};
        converged = pg_host->converged != 0;
    }
    if (inner_total) *inner_total += t;
    pg_backtracks += pg_host ? pg_host->backtracks : 0;
    return t;
}

template <typename T> void Solver<T>::subsolve(int which, const nmfx_opts &o, nmfx_result *out) {
    require_ready();
    HIP_TRY(hipSetDevice(device));
    std::memset(out, 0, sizeof *out);
    pg_backtracks = 0;
    long long inner = 0;
    if (which == 0) {                                                      // alspgrad_updateh! (:69-84)
        wt_times(W[wcur].p, X.p, true, nullptr);                           // set_w! (:63-67)
        out->niters = pg_subsolve(true, H[hcur].p, gramW_p, numH_p, o.maxsubiter, o.traceiter, (T)o.tolg, (T)o.beta, (T)o.sigma, &inner);
    } else {                                                               // alspgrad_updatew! (:225-240)
        times_ht(X.p, H[hcur].p, true, nullptr);                           // set_h! (:218-222)
        allreduce_w_side(false, nullptr);
        out->niters = pg_subsolve(false, W[wcur].p, gramH_p, numW_p, o.maxsubiter, o.traceiter, (T)o.tolg, (T)o.beta, (T)o.sigma, &inner);
    }
    HIP_TRY(hipStreamSynchronize(stream));
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
    long long inner = 0;
    T tolg = (T)o.tolg;                                                    // fresh ALSPGradUpd per solve! (:381-383)
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
        times_ht(X.p, Hc, true, nullptr);                                  // set_h! (:415)
        allreduce_w_side(false, nullptr);
        const long long itW = pg_subsolve(false, Wc, gramH_p, numW_p, o.maxsubiter, o.traceiter, tolg, (T)o.beta, (T)o.sigma, &inner);
        if (itW == 1) tolg = (T)((double)tolg * 0.1);                      // :419-421
        if (o.update_H) {
            stats_h(Hc, preH, nullptr);
            if (nranks > 1) RCCL_TRY(ncclAllReduce(hstat.p, hstat.p, (size_t)2 * K, ncclDouble, ncclSum, comm, stream));
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
