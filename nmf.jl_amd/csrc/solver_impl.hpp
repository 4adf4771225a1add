// solver_impl.hpp -- the per-algorithm kernel sequences and the iteration loop.
#pragma once
#include "solver.hpp"

namespace nmfx {

// ---------------------------------------------------------------------------
// objective without materialising WH:  D(r=j, c=i) = sum_a H(a,j) W(i,a), fused with the
// reduction over (X - WH)^2 or the KL term.   evaluate_objv: src/multupd.jl:81,148;
// src/projals.jl:65-74; src/alspgrad.jl:398
// ---------------------------------------------------------------------------
template <typename T>
void Solver<T>::enqueue_objective(int alg, const nmfx_opts &o, double *dst, const int *done) {
    const T *Wp = W[wcur].p, *Hp = H[hcur].p;
    const double bytes = (double)(P * N) * sizeof(T);
    if (alg == NMFX_ALG_MULTDIV) {
        EpiObjective<T, 1> e{X.p, P, obj_part.p, 0.0};
        gemm_wh("gemm_WH_kldiv", Hp, Wp, e, done, bytes);
    } else {
        EpiObjective<T, 0> e{X.p, P, obj_part.p, 0.0};
        gemm_wh("gemm_WH_sqdist", Hp, Wp, e, done, bytes);
    }
    const int nblk = last_blocks;   // one Float64 partial per block of the launch above
    int nextra = 0;
    if (alg == NMFX_ALG_PROJALS) {   // + 0.5*lambda_w*||W||^2 + 0.5*lambda_h*||H||^2   (projals.jl:67-72)
        const int nb = 1024;
        if (o.lambda_w > 0) {
            hipLaunchKernelGGL(sumsq_kernel<T>, dim3(nb), dim3(256), 0, stream, Wp, (int64_t)P * K, obj_part.p + nblk, done);
            hipLaunchKernelGGL(finish_sumsq_kernel<T>, dim3(1), dim3(64), 0, stream, obj_part.p + nblk, nb,
                               (T)((T)0.5 * (T)o.lambda_w), obj_extra.p, nextra, done);
            ++nextra;
        }
        if (o.lambda_h > 0) {
            hipLaunchKernelGGL(sumsq_kernel<T>, dim3(nb), dim3(256), 0, stream, Hp, (int64_t)K * N, obj_part.p + nblk + nb, done);
            hipLaunchKernelGGL(finish_sumsq_kernel<T>, dim3(1), dim3(64), 0, stream, obj_part.p + nblk + nb, nb,
                               (T)((T)0.5 * (T)o.lambda_h), obj_extra.p, nextra, done);
            ++nextra;
        }
    }
    if (alg == NMFX_ALG_GREEDYCD) {   // + lambda_w*norm(W,1) + lambda_h*norm(H,1)   (greedycd.jl:81-86)
        const int nb = 1024;
        if (o.lambda_w > 0) {
            hipLaunchKernelGGL(sumabs_kernel<T>, dim3(nb), dim3(256), 0, stream, Wp, (int64_t)P * K, obj_part.p + nblk, done);
            hipLaunchKernelGGL(finish_sumabs_kernel<T>, dim3(1), dim3(64), 0, stream, obj_part.p + nblk, nb, (T)o.lambda_w, obj_extra.p,
                               nextra, done);
            ++nextra;
        }
        if (o.lambda_h > 0) {
            hipLaunchKernelGGL(sumabs_kernel<T>, dim3(nb), dim3(256), 0, stream, Hp, (int64_t)K * N, obj_part.p + nblk + nb, done);
            hipLaunchKernelGGL(finish_sumabs_kernel<T>, dim3(1), dim3(64), 0, stream, obj_part.p + nblk + nb, nb, (T)o.lambda_h,
                               obj_extra.p, nextra, done);
            ++nextra;
        }
    }
    if (sharded()) {
        // the data term is a sum over column shards; regularisers: ||W||^2 is replicated, ||H||^2 is sharded.
        // Reduce the per-block partials to one value first, all-reduce it, then finish.
        hipLaunchKernelGGL(finish_objective_kernel<double>, dim3(1), dim3(256), 0, stream, obj_part.p, nblk, 1,
                           (const double *)nullptr, 0, obj_part.p, done);
        comm->all_reduce(obj_part.p, 1, CT_F64, false, stream);
        if ((alg == NMFX_ALG_PROJALS || alg == NMFX_ALG_GREEDYCD) && o.lambda_h > 0) {
            // sharded ||H||^2 term: sum the already-scaled shard terms (norm in T per shard; documented deviation)
            const int slot = (o.lambda_w > 0) ? 1 : 0;
            comm->all_reduce(obj_extra.p + slot, 1, CT_F64, false, stream);
        }
        hipLaunchKernelGGL(finish_objective_kernel<T>, dim3(1), dim3(256), 0, stream, obj_part.p, 1,
                           alg == NMFX_ALG_MULTDIV ? 1 : 0, obj_extra.p, nextra, dst, done);
    } else {
        hipLaunchKernelGGL(finish_objective_kernel<T>, dim3(1), dim3(256), 0, stream, obj_part.p, nblk,
                           alg == NMFX_ALG_MULTDIV ? 1 : 0, obj_extra.p, nextra, dst, done);
    }
    HIP_TRY(hipGetLastError());
}

template <typename T> void Solver<T>::allreduce_w_side(bool with_hstat, const int *done) {
    (void)done;
    if (!sharded()) return;
    timed("comm_allreduce_pack", 0.0, (double)(P * K + K * K) * sizeof(T), [&] {
        comm->group_start();
        comm->all_reduce(pack.p, pack.count, CT, false, stream);
        if (with_hstat) comm->all_reduce(hstat.p, (size_t)2 * K, CT_F64, false, stream);
        comm->group_end();
    });
}

// Row-sharded W side, step 1: the blocked numerator partial sums (times_ht with w_blocked) are reduce-scattered by row
// blocks; the small tail of the packed buffer [ H_g H_g' | rowsum(H_g) ] and the H statistics are all-reduced in the same
// group.  The rank's rows of the summed numerator land back in numW (standard layout, ld P) at [row0, row0 + Pc).
template <typename T> void Solver<T>::scatter_w_numerator(bool with_hstat, const int *done, bool with_tail) {
    timed("comm_reduce_scatter_numW", 0.0, (double)(P * K + K * K) * sizeof(T), [&] {
        comm->group_start();
        comm->reduce_scatter(numW_p, rs_out.p, (size_t)Pc * K, CT, stream);
        if (with_tail) comm->all_reduce(gramH_p, (size_t)K * K + (size_t)K, CT, false, stream);
        if (with_hstat) comm->all_reduce(hstat.p, (size_t)2 * K, CT_F64, false, stream);
        comm->group_end();
    });
    hipLaunchKernelGGL(piece_to_rows_kernel<T>, dim3(flat_grid(Pc * K)), dim3(256), 0, stream, numW_p, rs_out.p, P, K, Pc, row0, done);
    HIP_TRY(hipGetLastError());
}

// stop_condition statistics of this rank's rows of W (src/common.jl:95-99): per-column partial sums over [row0, row0 + Pc),
// written into the tail of the all-gather chunk; gather_w_rows adds the ranks' partials in rank order.
template <typename T> void Solver<T>::stats_w_rows(const T *Wn, const T *Wo, const int *done) {
    const int chunks = (int)std::max<int64_t>(1, std::min<int64_t>(64, Pc / 1024));
    timed("stats_W", 0.0, 2.0 * Pc * K * sizeof(T), [&] {
        hipLaunchKernelGGL(col_stats_kernel<T>, dim3(chunks, (unsigned)K), dim3(256), 0, stream, Wn + row0, Wo + row0, Pc, P,
                           (int)K, stat_part.p, done);
        hipLaunchKernelGGL(finalize_partials_kernel<double>, dim3((unsigned)((2 * K + 3) / 4)), dim3(256), 0, stream,
                           stat_part.p, chunks, (int)(2 * K), (int)(2 * K),
                           reinterpret_cast<double *>(ag_send.p + (size_t)Pc * K * sizeof(T)), done);
        HIP_TRY(hipGetLastError());
    });
}

// Row-sharded W side, step 2: every rank contributes its Pc updated rows (+ its column statistics partials), the
// all-gather re-assembles the full W on every rank (identical bits everywhere) and wstat = sum of the partials.
template <typename T> void Solver<T>::gather_w_rows(T *Wfull, bool with_stats, const int *done) {
    hipLaunchKernelGGL(rows_to_piece_kernel<T>, dim3(flat_grid(Pc * K)), dim3(256), 0, stream, reinterpret_cast<T *>(ag_send.p), Wfull,
                       P, K, Pc, row0, done);
    timed("comm_all_gather_W", 0.0, (double)(P * K) * sizeof(T), [&] {
        comm->all_gather(ag_send.p, ag_recv.p, ag_chunk_bytes, CT_BYTE, stream);
    });
    hipLaunchKernelGGL(gathered_to_full_kernel<T>, dim3(flat_grid(P * K)), dim3(256), 0, stream, Wfull, ag_recv.p, nranks, ag_chunk_bytes,
                       P, K, Pc, (int64_t)0, with_stats ? (int)(2 * K) : 0, with_stats ? wstat.p : (double *)nullptr, 0, done);
    HIP_TRY(hipGetLastError());
}

// ---------------------------------------------------------------------------
// MultUpdate, MSE objective.  update_wh!(::MultUpdMSE), src/multupd.jl:83-116, in Gram form:
//   H <- H .* max(0, W'X - lh) ./ ((W'W) H + delta)      [reference: W'(WH)]
//   W <- W .* max(0, XH' - lw) ./ (W (HH') + delta)      [reference: (WH)H']
// ---------------------------------------------------------------------------
template <typename T> void Solver<T>::enqueue_multmse(const nmfx_opts &o, long long t) {
    (void)t;
    if (smallk_ok()) { enqueue_multmse_smallk(o, t); return; }
    const int *done = done_flag();
    if (o.update_H) {
        const T *Wp = W[wcur].p;
        const T *Ho = H[hcur].p;
        T *Hn = H[hcur ^ 1].p;
        // :98 W'X (and W'W, same launch) stay as split-K slabs: the Gram slabs get a tiny reduction, the update GEMM sums
        // the numerator slabs in its epilogue (ascending slab order, like reduce_slabs_kernel), and it also
        // produces the stop_condition statistics of H -- three launches for the whole H phase.
        const bool fusedrs = rs_fused();
        // row-sharded fused step: from the second iteration on W'W arrives all-reduced from the ranks' own row blocks (computed
        // behind the W update, multmse_w_rows_fused) instead of being recomputed over all P rows on every rank inside this launch
        const bool have_gram = fusedrs && gramw_sharded_valid;
        h_reduce_pair = fusedrs;
        if (w_res_blocked && have_gram) { wt_blocked = reinterpret_cast<const T *>(Wblk[wb].p); wt_blk_stride = (int64_t)(blk_chunk / sizeof(T)); }
        else w_sync(done);
        // (row-sharded fused step, Gram already all-reduced: up to 8 split-K slabs of the numerator go straight into the update's epilogue)
        const bool many = have_gram && s_h > 2 && s_h <= 8;
        h_keep_max = many ? 8 : 2;
        wt_times(Wp, X.p, !have_gram, done, /*keep_slabs=*/true);
        h_keep_max = 2;
        wt_blocked = nullptr;
        h_reduce_pair = false;
        const bool eight = h_in_slabs && h_nslab > 2;
        if constexpr (sizeof(T) == 4) {
            if (ht_active && !eight) {   // the new H also transposed, for the X*H' product below
                EpiMultUpdate<T, 2> e{h_num(), h_in_slabs ? h_nslab : 1, h_stride, Ho, Hn, K, (T)o.lambda_h, (T)o.delta, stat_part.p, (int)K};  // :99-103
                e.outT = Ht[hcur ^ 1].p; e.ldT = N;
                gemm<KCONTIG, KCONTIG>("gemm_WtWH_updH", Ho, K, N, gramW_p, K, K, K, 1, true, e, done, (4.0 + e.nslab) * K * N * sizeof(T));
            } else if (ht_active) {
                EpiMultUpdate<T, 2, 8> e{h_num(), h_nslab, h_stride, Ho, Hn, K, (T)o.lambda_h, (T)o.delta, stat_part.p, (int)K};
                e.outT = Ht[hcur ^ 1].p; e.ldT = N;
                gemm<KCONTIG, KCONTIG>("gemm_WtWH_updH", Ho, K, N, gramW_p, K, K, K, 1, true, e, done, (4.0 + e.nslab) * K * N * sizeof(T));
            }
        }
        if (!ht_active && eight) {
            EpiMultUpdate<T, 1, 8> e{h_num(), h_nslab, h_stride, Ho, Hn, K, (T)o.lambda_h, (T)o.delta, stat_part.p, (int)K};
            gemm<KCONTIG, KCONTIG>("gemm_WtWH_updH", Ho, K, N, gramW_p, K, K, K, 1, true, e, done, (3.0 + e.nslab) * K * N * sizeof(T));
        } else if (!ht_active) {
        EpiMultUpdate<T, 1> e{h_num(), h_num_nslab(), h_stride, Ho, Hn, K, (T)o.lambda_h, (T)o.delta, stat_part.p, (int)K};  // :99-103
        gemm<KCONTIG, KCONTIG>("gemm_WtWH_updH", Ho, K, N, gramW_p, K, K, K, 1, true, e, done, (3.0 + h_num_nslab()) * K * N * sizeof(T));
        }
        h_stat_chunks = last_tiles_r;
        // (fused row-sharded step: finalised by the W side's combine launch; one GPU, nothing tracked: by the launch behind the W update)
        if (!fusedrs && !stats_fuse_ok(o)) stats_h_finalize(last_tiles_r, done);
        hcur ^= 1;
    }
    const T *Hp = H[hcur].p;
    const T *Wo = W[wcur].p;
    T *Wn = W[wcur ^ 1].p;
    if (rs_fused()) { multmse_w_rows_fused(o, t); return; }
    const T *HtP = ht_active ? Ht[hcur].p : nullptr;
    if (row_sharded()) {
        // sharded: X_g H_g' partial sums -> reduce-scatter by row blocks -> this rank updates ITS Pc rows -> all-gather
        w_blocked = true;
        times_ht(X.p, Hp, true, done, false, HtP);
        w_blocked = false;
        scatter_w_numerator(o.update_H != 0, done);
        EpiMultUpdate<T, 0> e{numW_p + row0, 1, 0, Wo + row0, Wn + row0, P, (T)o.lambda_w, (T)o.delta, nullptr, 0};   // :110-114
        gemm<KSTRIDED, KSTRIDED>("gemm_WHHt_updW", gramH_p, K, K, Wo + row0, P, Pc, K, 1, false, e, done, 3.0 * Pc * K * sizeof(T));
        stats_w_rows(Wn, Wo, done);
        gather_w_rows(Wn, true, done);
        wcur ^= 1;
        return;
    }
    // :109 XH': single GPU -> slabs are consumed by the update GEMM's epilogue; replicated-W mode -> reduce into the packed
    // buffer first, because the all-reduce needs the rank-local sum
    const bool w_slabs = !sharded();
    times_ht(X.p, Hp, true, done, /*keep_slabs=*/w_slabs, HtP);
    allreduce_w_side(o.update_H != 0, done);
    EpiMultUpdate<T, 0> e{w_num(), w_num_nslab(), w_stride, Wo, Wn, P, (T)o.lambda_w, (T)o.delta, nullptr, 0};   // :110-114
    gemm<KSTRIDED, KSTRIDED>("gemm_WHHt_updW", gramH_p, K, K, Wo, P, P, K, 1, false, e, done, 3.0 * P * K * sizeof(T));
    if (stats_fuse_ok(o)) stats_w_check_fused(Wn, Wo, o, t, done);
    else stats_w(Wn, Wo, done);
    wcur ^= 1;
}

// The row-sharded W side of MultUpdate-MSE with everything around the two collectives fused (8 launches per iteration instead of
// 17 at the 8-rank shard shape, where an iteration is 0.4 ms and every small launch is 6-17 us):
//   X_g H_g' (+ H_g H_g' tail pieces)  ->  ONE combine launch (numerator slabs into the blocked send buffer, Gram pieces, H statistics)
//   -> reduce-scatter + the small all-reduces  ->  update GEMM on the rank's rows: numerator straight from the reduce-scatter's
//   output, new rows straight into the rank's chunk of the all-gather buffer (EpiMultUpdateRows)  ->  IN-PLACE all-gather
//   ->  one launch that unpacks W and takes stop_condition's column sums over all rows (gather_stats_kernel), one block that adds them
//   up and runs the stop rule (stats_check_kernel).
template <typename T> void Solver<T>::multmse_w_rows_fused(const nmfx_opts &o, long long t) {
    if (PeerComm *pc = peer()) {
        const size_t need = (size_t)Pc * K * sizeof(T) + 2 * ((size_t)K * K * sizeof(T) + 256) + (size_t)2 * K * sizeof(double) + 1024;   // (the pull form's larger chunk: blocked_residency_ok)
        if (nranks <= EPI_MAX_PIECES && need <= pc->slot_bytes) { multmse_w_rows_fused_peer(o, t, pc); return; }
    }
    const int *done = done_flag();
    const T *Hp = H[hcur].p;
    const T *Wo = W[wcur].p;
    T *Wn = W[wcur ^ 1].p;
    w_blocked = true; w_defer_combine = true;
    times_ht(X.p, Hp, true, done, false, ht_active ? Ht[hcur].p : nullptr);
    w_blocked = false; w_defer_combine = false;
    const unsigned nb1 = w_direct ? 0u : (unsigned)(P / 256 * K), nb2 = (unsigned)((K * K + 63) / 64), nb3 = o.update_H ? (unsigned)((2 * K + 3) / 4) : 0u;
    const bool fuse_check = o.track_objective == 0 && o.stop_sums == 0;
    const bool defer = fuse_check && defer_enabled && blocked_residency_ok();   // the stop rule of this iteration rides in the NEXT combine launch
    double *hs = defer ? hstat_of(t) : hstat.p;
    timed("combine_W", 0.0, ((double)P * K * (w_nslab + 1) + (double)K * K * (w_pieces + 1)) * sizeof(T), [&] {
        const unsigned nbd = defer_pending ? STAT_BLOCKS : 0u;
        hipLaunchKernelGGL(w_side_combine_kernel<T>, dim3(nb1 + nb2 + nb3 + nbd), dim3(256), 0, stream, numW_p, slabs.p + slab_w_off, P, K, Pc, w_nslab,
                           w_stride, gramH_p, slabs.p + gram_slab_off, (int64_t)K * K, w_pieces, (int64_t)K * K, stat_part.p, h_stat_chunks,
                           (int)(2 * K), hs, nb1, nb2, done, CombineDst<T>{{}, {}, {}, 0}, defer_args(nbd));
        HIP_TRY(hipGetLastError());
        defer_pending = false;
    });
    timed("comm_reduce_scatter_numW", 0.0, (double)(P * K + K * K) * sizeof(T), [&] {
        comm->group_start();
        comm->reduce_scatter(numW_p, rs_out.p, (size_t)Pc * K, CT, stream);
        comm->all_reduce(gramH_p, (size_t)K * K, CT, false, stream);
        if (o.update_H) comm->all_reduce(hs, (size_t)2 * K, CT_F64, false, stream);
        comm->group_end();
    });
    const int cpp = (int)std::max<int64_t>(1, std::min<int64_t>(64 / nranks, Pc / 1024));   // statistics chunks per row block
    if (blocked_residency_ok()) {
        // ---- W stays in the all-gather's layout between iterations (solver.hpp: Wblk): no unpack launch ----------------------------
        blk_cpp = cpp;
        blk_chunk = ((size_t)Pc * K * sizeof(T) + (size_t)cpp * 2 * K * sizeof(double) + 255) / 256 * 256;
        for (auto &b : Wblk) b.ensure(blk_chunk * (size_t)nranks);
        const int64_t blk_el = (int64_t)(blk_chunk / sizeof(T));
        const T *Wo_own = w_res_blocked ? reinterpret_cast<const T *>(Wblk[wb].p) + (int64_t)rank * blk_el : Wo + row0;
        const int64_t ldo = w_res_blocked ? Pc : P;
        unsigned char *mine_b = Wblk[wb ^ 1].p + (size_t)rank * blk_chunk;
        T *mine = reinterpret_cast<T *>(mine_b);
        double *tail = reinterpret_cast<double *>(mine_b + (size_t)Pc * K * sizeof(T));
        EpiMultUpdateRows<T> e{rs_out.p, Pc, Wo_own, ldo, mine, (T)o.lambda_w, (T)o.delta};               // multupd.jl:110-114
        gemm<KSTRIDED, KSTRIDED>("gemm_WHHt_updW", gramH_p, K, K, Wo_own, ldo, Pc, K, 1, false, e, done, 3.0 * Pc * K * sizeof(T));
        int sg = 0;
        if (o.update_H) {
            sg = pick_splits((int)((K / 64) * (K / 64)), Pc);
            EpiStore<T> eg{slabs.p + gram_slab_off, K, (int64_t)K * K, nullptr};
            force_quarter_tiles = true;
            gemm<KCONTIG, KCONTIG>("gemm_WtW_rows", mine, Pc, K, mine, Pc, K, Pc, sg, true, eg, done, (double)(Pc * K) * sizeof(T));
            force_quarter_tiles = false;
            // (16 or more slabs keep their own combine launch: reduce_many_slabs_kernel adds them in another -- fixed -- order, and the peer
            // transport's step, which must give the same bits, uses it)
            if (sg >= 16) { reduce_slabs_from("reduce_WtW", gramW_p, slabs.p + gram_slab_off, (int64_t)K * K, sg, done); sg = 0; }
        }
        // ONE launch: stop_condition's sums over the rank's OWN rows (the chunking and arithmetic of gather_stats_kernel) into the tail of
        // the chunk, and the split-K combine of the own-rows Gram (reduce_slabs_vec_kernel's arithmetic)
        timed("stats_W_rows+reduce_WtW", 0.0, (2.0 * Pc * K + (double)K * K * (sg + 1)) * sizeof(T), [&] {
            constexpr int V = 16 / (int)sizeof(T);
            const int64_t gnvec = (int64_t)K * K / V;
            const unsigned nbs = (unsigned)(cpp * K), nbr = sg > 0 ? (unsigned)((gnvec + 255) / 256) : 0u;
            hipLaunchKernelGGL(rows_tail_kernel<T>, dim3(nbs + nbr), dim3(256), 0, stream, Wo_own - (int64_t)rank * (w_res_blocked ? blk_el : Pc), Wblk[wb ^ 1].p, blk_chunk,
                               P, Pc, cpp, (int)K, tail, w_res_blocked ? blk_el : Pc, ldo, rank, nbs, gramW_p, slabs.p + gram_slab_off, gnvec, sg, (int64_t)K * K, done);
            HIP_TRY(hipGetLastError());
        });
        timed("comm_all_gather_W", 0.0, (double)(P * K) * sizeof(T), [&] {
            comm->group_start();
            comm->all_gather(mine_b, Wblk[wb ^ 1].p, blk_chunk, CT_BYTE, stream);
            if (o.update_H) comm->all_reduce(gramW_p, (size_t)K * K, CT, false, stream);
            comm->group_end();
        });
        gramw_sharded_valid = o.update_H != 0;
        if (defer) {
            defer_pending = true; defer_t = t; defer_opts = o;
        } else {
        timed("stats_check", 0.0, 0.0, [&] {
            hipLaunchKernelGGL(stats_check_kernel<T>, dim3(STAT_BLOCKS), dim3(256), 0, stream, reinterpret_cast<const double *>(Wblk[wb ^ 1].p + (size_t)Pc * K * sizeof(T)), nranks * cpp, (int)K,
                               wstat.p, ctrl, o.update_H ? hs : (const double *)nullptr, (int)k, (T)o.tol, t, fuse_check ? 1 : 0, done, cpp,
                               (int64_t)(blk_chunk / sizeof(double)), stat_ticket());
            HIP_TRY(hipGetLastError());
        });
        }
        check_fused = fuse_check;
        wb ^= 1;
        w_res_blocked = true;
        w_std_stale = true;
        wcur ^= 1;
        return;
    }
    w_sync(done);
    w_res_blocked = false;
    const size_t chunk = (size_t)Pc * K * sizeof(T);
    T *mine = reinterpret_cast<T *>(ag_recv.p + (size_t)rank * chunk);
    EpiMultUpdateRows<T> e{rs_out.p, Pc, Wo + row0, P, mine, (T)o.lambda_w, (T)o.delta};                  // multupd.jl:110-114
    gemm<KSTRIDED, KSTRIDED>("gemm_WHHt_updW", gramH_p, K, K, Wo + row0, P, Pc, K, 1, false, e, done, 3.0 * Pc * K * sizeof(T));
    if (o.update_H) {
        // W'W for the next H update from the rank's OWN new rows (2 Pc k^2 instead of 2 p k^2 on every rank), summed over the ranks
        // by a k x k all-reduce that travels in the same group as the all-gather
        const int sg = pick_splits((int)((K / 64) * (K / 64)), Pc);
        EpiStore<T> eg{slabs.p + gram_slab_off, K, (int64_t)K * K, nullptr};
        force_quarter_tiles = true;
        gemm<KCONTIG, KCONTIG>("gemm_WtW_rows", mine, Pc, K, mine, Pc, K, Pc, sg, true, eg, done, (double)(Pc * K) * sizeof(T));
        force_quarter_tiles = false;
        reduce_slabs_from("reduce_WtW", gramW_p, slabs.p + gram_slab_off, (int64_t)K * K, sg, done);
    }
    timed("comm_all_gather_W", 0.0, (double)(P * K) * sizeof(T), [&] {
        comm->group_start();
        comm->all_gather(mine, ag_recv.p, chunk, CT_BYTE, stream);
        if (o.update_H) comm->all_reduce(gramW_p, (size_t)K * K, CT, false, stream);
        comm->group_end();
    });
    gramw_sharded_valid = o.update_H != 0;
    timed("gather_W_stats", 0.0, 3.0 * P * K * sizeof(T), [&] {
        hipLaunchKernelGGL(gather_stats_kernel<T>, dim3((unsigned)(nranks * cpp), (unsigned)K), dim3(256), 0, stream, Wn, Wo, ag_recv.p, chunk, P, Pc, cpp, (int)K,
                           stat_part.p, done);
        hipLaunchKernelGGL(stats_check_kernel<T>, dim3(1), dim3(256), 0, stream, stat_part.p, nranks * cpp, (int)K, wstat.p, ctrl,
                           o.update_H ? hs : (const double *)nullptr, (int)k, (T)o.tol, t, fuse_check ? 1 : 0, done);
        HIP_TRY(hipGetLastError());
    });
    check_fused = fuse_check;
    wcur ^= 1;
}

// The same step on the peer-to-peer transport (peer.hpp): no collective launches at all.
//   exchange 1 : the X_g H_g' launch stores row block g of the numerator STRAIGHT into rank g's receive slot (EpiStorePeer; with split-K
//                slabs: the combine launch does), the combine launch stores H_g H_g' and the H statistics into every rank's slot
//                -> flag -> one-block wait -> ONE launch adds the ranks' contributions in rank order (peer_sum3_kernel)
//   exchange 2 : the rank's new rows and W_g'W_g pushed into every rank's slot by one copy launch each -> flag -> wait ->
//                gather_stats_kernel reads the ranks' row blocks straight out of the slots
// Same arithmetic in the same order as the in-process group's reduction kernels: bit-identical iterates (tests/test_gpu_peer.py).
template <typename T> void Solver<T>::multmse_w_rows_fused_peer(const nmfx_opts &o, long long t, PeerComm *pc) {
    const int *done = done_flag();
    const T *Hp = H[hcur].p;
    const T *Wo = W[wcur].p;
    T *Wn = W[wcur ^ 1].p;
    const int G = nranks;
    const size_t piece_b = (size_t)Pc * K * sizeof(T), gram_b = (size_t)K * K * sizeof(T), hs_b = (size_t)2 * K * sizeof(double);
    // ---- exchange 1
    pc->group_start();
    pc->group_stream = stream;
    const size_t o_num = pc->direct_reserve(piece_b), o_gram = pc->direct_reserve(gram_b), o_hs = o.update_H ? pc->direct_reserve(hs_b) : 0;
    const bool fuse_check0 = o.track_objective == 0 && o.stop_sums == 0;
    const bool defer = fuse_check0 && defer_enabled && blocked_residency_ok();   // the stop rule of this iteration rides in the NEXT combine launch
    double *hs = defer ? hstat_of(t) : hstat.p;
    CombineDst<T> pd;
    std::memset(&pd, 0, sizeof pd);
    pd.n = G;
    for (int q = 0; q < G; ++q) {
        pd.num[q] = reinterpret_cast<T *>(pc->direct_dst(q, o_num));
        pd.gram[q] = reinterpret_cast<T *>(pc->direct_dst(q, o_gram));
        pd.hstat[q] = reinterpret_cast<double *>(pc->direct_dst(q, o_hs));
    }
    const unsigned char *s_num = pc->direct_src(0, o_num), *s_gram = pc->direct_src(0, o_gram), *s_hs = pc->direct_src(0, o_hs);
    w_blocked = true; w_defer_combine = true; peer_dst = &pd;
    times_ht(X.p, Hp, true, done, false, ht_active ? Ht[hcur].p : nullptr);
    w_blocked = false; w_defer_combine = false; peer_dst = nullptr;
    {
        const unsigned nb1 = w_direct ? 0u : (unsigned)(P / 256 * K), nb2 = (unsigned)((K * K + 63) / 64), nb3 = o.update_H ? (unsigned)((2 * K + 3) / 4) : 0u;
        timed("combine_W", 0.0, ((double)P * K * (w_nslab + 1) + (double)K * K * (w_pieces + 1)) * sizeof(T), [&] {
            const unsigned nbd = defer_pending ? STAT_BLOCKS : 0u;
            hipLaunchKernelGGL(w_side_combine_kernel<T>, dim3(nb1 + nb2 + nb3 + nbd), dim3(256), 0, stream, numW_p, slabs.p + slab_w_off, P, K, Pc, w_nslab,
                               w_stride, gramH_p, slabs.p + gram_slab_off, (int64_t)K * K, w_pieces, (int64_t)K * K, stat_part.p, h_stat_chunks,
                               (int)(2 * K), hs, nb1, nb2, done, pd, defer_args(nbd));
            HIP_TRY(hipGetLastError());
            defer_pending = false;
        });
    }
    timed("comm_p2p_flag_wait_numW", 0.0, (double)(P * K + K * K) * sizeof(T), [&] { pc->group_end(); });
    {
        const unsigned nb1 = (unsigned)std::min<int64_t>((Pc * K + 1023) / 1024, 2048), nb2 = (unsigned)std::min<int64_t>((K * K + 255) / 256, 256), nb3 = o.update_H ? 2u : 0u;
        timed("sum_numW_slots", 0.0, (double)G * (Pc * K + K * K) * sizeof(T), [&] {
            hipLaunchKernelGGL(peer_sum3_kernel<T>, dim3(nb1 + nb2 + nb3), dim3(256), 0, stream, rs_out.p, s_num, (size_t)Pc * K, gramH_p, s_gram, (size_t)K * K, hs,
                               s_hs, o.update_H ? (size_t)2 * K : (size_t)0, pc->slot_bytes, G, nb1, nb2, done);
            HIP_TRY(hipGetLastError());
        });
    }
    if (blocked_residency_ok()) {
        // ---- exchange 2 as a PULL, W resident in the all-gather's layout (round 6) ---------------------------------------------------
        // The rank's update writes its new rows twice -- into its chunk of the blocked W buffer (cached: its own next products read them
        // there) and into ITS OWN window slot (write-through) --, the own-rows statistics and the own-rows Gram follow them into the
        // window, then flag -> wait -> ONE launch that copies the peers' chunks out of their windows into the blocked buffer, adds the
        // ranks' Grams in rank order and runs the stop rule on the ranks' statistics (peer_pull_kernel).  Against the push form below:
        // no 16 MB of uncached stores (42 us at the 8-rank shard shape), no unpack of 16 MB out of the own window (25 us), no separate
        // one-block stop-rule launch, and the next W'X contracts over the blocked buffer in place.  Same arithmetic in the same order:
        // bit-identical to the push form and to the in-process group (tests/test_gpu_peer.py).
        const int cpp = (int)std::max<int64_t>(1, std::min<int64_t>(64 / nranks, Pc / 1024));   // statistics chunks per row block
        const bool fuse_check = o.track_objective == 0 && o.stop_sums == 0;
        blk_cpp = cpp;
        blk_chunk = ((size_t)Pc * K * sizeof(T) + (size_t)cpp * 2 * K * sizeof(double) + 255) / 256 * 256;
        for (auto &b : Wblk) b.ensure(blk_chunk * (size_t)nranks);
        const int64_t blk_el = (int64_t)(blk_chunk / sizeof(T));
        const T *Wo_own = w_res_blocked ? reinterpret_cast<const T *>(Wblk[wb].p) + (int64_t)rank * blk_el : Wo + row0;
        const int64_t ldo = w_res_blocked ? Pc : P;
        unsigned char *mine_b = Wblk[wb ^ 1].p + (size_t)rank * blk_chunk;
        T *mine = reinterpret_cast<T *>(mine_b);
        double *tail = reinterpret_cast<double *>(mine_b + (size_t)Pc * K * sizeof(T));
        pc->group_start();
        pc->group_stream = stream;
        const size_t o_w = pc->direct_reserve(blk_chunk), o_gw = o.update_H ? pc->direct_reserve(gram_b) : 0;
        unsigned char *pub = pc->direct_dst(rank, o_w);                       // this rank's slot in its own window
        T *pub_gram = reinterpret_cast<T *>(pc->direct_dst(rank, o_gw));
        PullSrc ps;
        std::memset(&ps, 0, sizeof ps);
        for (int q = 0; q < G; ++q) { ps.chunk[q] = pc->pull_src(q, o_w); ps.gram[q] = pc->pull_src(q, o_gw); }
        EpiMultUpdateRows<T> e{rs_out.p, Pc, Wo_own, ldo, mine, (T)o.lambda_w, (T)o.delta};               // multupd.jl:110-114
        e.out2 = reinterpret_cast<T *>(pub);
        gemm<KSTRIDED, KSTRIDED>("gemm_WHHt_updW", gramH_p, K, K, Wo_own, ldo, Pc, K, 1, false, e, done, 3.0 * Pc * K * sizeof(T));
        int sg = 0;
        bool gram_pub = false;
        if (o.update_H) {
            sg = pick_splits((int)((K / 64) * (K / 64)), Pc);
            EpiStore<T> eg{slabs.p + gram_slab_off, K, (int64_t)K * K, nullptr};
            force_quarter_tiles = true;
            gemm<KCONTIG, KCONTIG>("gemm_WtW_rows", mine, Pc, K, mine, Pc, K, Pc, sg, true, eg, done, (double)(Pc * K) * sizeof(T));
            force_quarter_tiles = false;
            // (16 or more slabs keep their own combine launch -- reduce_many_slabs_kernel's fixed order, like every other form of this step)
            if (sg >= 16) { reduce_slabs_from("reduce_WtW", gramW_p, slabs.p + gram_slab_off, (int64_t)K * K, sg, done); sg = 0; gram_pub = true; }
        }
        timed("stats_W_rows+reduce_WtW", 0.0, (2.0 * Pc * K + (double)K * K * (sg + 1)) * sizeof(T), [&] {
            constexpr int V = 16 / (int)sizeof(T);
            const int64_t gnvec = (int64_t)K * K / V;
            const unsigned nbs = (unsigned)(cpp * K), nbr = sg > 0 ? (unsigned)((gnvec + 255) / 256) : 0u;
            hipLaunchKernelGGL(rows_tail_kernel<T>, dim3(nbs + nbr), dim3(256), 0, stream, Wo_own - (int64_t)rank * (w_res_blocked ? blk_el : Pc), Wblk[wb ^ 1].p, blk_chunk,
                               P, Pc, cpp, (int)K, tail, w_res_blocked ? blk_el : Pc, ldo, rank, nbs, gramW_p, slabs.p + gram_slab_off, gnvec, sg, (int64_t)K * K, done,
                               reinterpret_cast<double *>(pub + (size_t)Pc * K * sizeof(T)), pub_gram);
            if (gram_pub) {   // the Gram came out of its own combine launch: one more copy, into the own window only
                PeerWin self;
                for (auto &q : self.p) q = nullptr;
                self.p[0] = pc->mine;
                hipLaunchKernelGGL(peer_push_kernel, dim3(PeerComm::grid_for(gram_b / 16 + 1), 1), dim3(256), 0, stream, self, pc->direct_off(o_gw),
                                   reinterpret_cast<const unsigned char *>(gramW_p), gram_b, (size_t)0, 0);
            }
            HIP_TRY(hipGetLastError());
        });
        timed("comm_p2p_flag_wait_W", 0.0, (double)(P * K) * sizeof(T), [&] { pc->group_end(); });
        gramw_sharded_valid = o.update_H != 0;
        timed("pull_W_rows+stats_check", 0.0, ((double)(G - 1) * 2.0 * Pc * K + (double)G * K * K) * sizeof(T), [&] {
            const unsigned nbc = (unsigned)std::max<size_t>(1, std::min<size_t>((blk_chunk / 16 + 1023) / 1024, 128));   // blocks per peer chunk
            const unsigned nbg = o.update_H ? (unsigned)std::min<int64_t>(((int64_t)K * K * sizeof(T) / 16 + 255) / 256, 256) : 0u;
            // (deferred stop rule: no statistics blocks here -- the next combine launch, or defer_flush, runs the rule on the tails this launch
            // copies into the blocked buffer; the copy blocks then keep their `done` guard)
            hipLaunchKernelGGL(peer_pull_kernel<T>, dim3((unsigned)G * nbc + nbg + (defer ? 0u : STAT_BLOCKS)), dim3(256), 0, stream, ps, rank, G, Wblk[wb ^ 1].p, blk_chunk, nbc, gramW_p,
                               (size_t)K * K, nbg, (size_t)Pc * K * sizeof(T), cpp, (int)K, wstat.p, ctrl, o.update_H ? hs : (const double *)nullptr, (int)k,
                               (T)o.tol, t, fuse_check ? 1 : 0, (fuse_check && !defer) ? (const int *)nullptr : done, done, stat_ticket());
            HIP_TRY(hipGetLastError());
        });
        if (defer) { defer_pending = true; defer_t = t; defer_opts = o; }
        check_fused = fuse_check;
        wb ^= 1;
        w_res_blocked = true;
        w_std_stale = true;
        wcur ^= 1;
        return;
    }
    w_sync(done);
    w_res_blocked = false;
    const size_t chunk = piece_b;
    T *mine = reinterpret_cast<T *>(ag_recv.p + (size_t)rank * chunk);
    EpiMultUpdateRows<T> e{rs_out.p, Pc, Wo + row0, P, mine, (T)o.lambda_w, (T)o.delta};                  // multupd.jl:110-114
    gemm<KSTRIDED, KSTRIDED>("gemm_WHHt_updW", gramH_p, K, K, Wo + row0, P, Pc, K, 1, false, e, done, 3.0 * Pc * K * sizeof(T));
    if (o.update_H) {
        const int sg = pick_splits((int)((K / 64) * (K / 64)), Pc);
        EpiStore<T> eg{slabs.p + gram_slab_off, K, (int64_t)K * K, nullptr};
        force_quarter_tiles = true;
        gemm<KCONTIG, KCONTIG>("gemm_WtW_rows", mine, Pc, K, mine, Pc, K, Pc, sg, true, eg, done, (double)(Pc * K) * sizeof(T));
        force_quarter_tiles = false;
        reduce_slabs_from("reduce_WtW", gramW_p, slabs.p + gram_slab_off, (int64_t)K * K, sg, done);
    }
    // ---- exchange 2 (push form; NMFX_P2P_PULL=0 or shapes the blocked residency does not cover)
    pc->group_start();
    pc->group_stream = stream;
    const size_t o_w = pc->direct_reserve(piece_b), o_gw = o.update_H ? pc->direct_reserve(gram_b) : 0;
    const unsigned char *s_w = pc->direct_src(0, o_w), *s_gw = pc->direct_src(0, o_gw);
    timed("push_W_rows", 0.0, (double)G * (Pc * K) * sizeof(T), [&] {
        // (the pushes carry no `done` guard: behind the stop they re-send the last rows, which nobody reads)
        hipLaunchKernelGGL(peer_push_kernel, dim3(PeerComm::grid_for(piece_b / 16 + 1), G), dim3(256), 0, stream, pc->win, pc->direct_off(o_w), reinterpret_cast<const unsigned char *>(mine),
                           piece_b, (size_t)0, rank);
        if (o.update_H)
            hipLaunchKernelGGL(peer_push_kernel, dim3(PeerComm::grid_for(gram_b / 16 + 1), G), dim3(256), 0, stream, pc->win, pc->direct_off(o_gw),
                               reinterpret_cast<const unsigned char *>(gramW_p), gram_b, (size_t)0, rank);
        HIP_TRY(hipGetLastError());
    });
    timed("comm_p2p_flag_wait_W", 0.0, (double)(P * K) * sizeof(T), [&] { pc->group_end(); });
    gramw_sharded_valid = o.update_H != 0;
    const bool fuse_check = o.track_objective == 0;
    const int cpp = (int)std::max<int64_t>(1, std::min<int64_t>(64 / nranks, Pc / 1024));   // chunks per piece
    timed("gather_W_stats", 0.0, 3.0 * P * K * sizeof(T), [&] {
        hipLaunchKernelGGL(gather_stats_kernel<T>, dim3((unsigned)(nranks * cpp), (unsigned)K), dim3(256), 0, stream, Wn, Wo, s_w, pc->slot_bytes, P, Pc, cpp, (int)K,
                           stat_part.p, done);
        if (o.update_H)
            hipLaunchKernelGGL(peer_sum3_kernel<T>, dim3((unsigned)std::min<int64_t>((K * K + 255) / 256, 256)), dim3(256), 0, stream, gramW_p, s_gw, (size_t)K * K, (T *)nullptr,
                               (const unsigned char *)nullptr, (size_t)0, (double *)nullptr, (const unsigned char *)nullptr, (size_t)0, pc->slot_bytes, G,
                               (unsigned)std::min<int64_t>((K * K + 255) / 256, 256), 0u, done);
        hipLaunchKernelGGL(stats_check_kernel<T>, dim3(1), dim3(256), 0, stream, stat_part.p, nranks * cpp, (int)K, wstat.p, ctrl,
                           o.update_H ? hs : (const double *)nullptr, (int)k, (T)o.tol, t, fuse_check ? 1 : 0, done);
        HIP_TRY(hipGetLastError());
    });
    check_fused = fuse_check;
    wcur ^= 1;
}

// ---------------------------------------------------------------------------
// MultUpdate, divergence objective.  update_wh!(::MultUpdDiv), src/multupd.jl:150-193.
// Q = X ./ (WH + delta) is produced by the W*H GEMM's epilogue (WH itself never hits HBM).
// ---------------------------------------------------------------------------
template <typename T> void Solver<T>::enqueue_multdiv(const nmfx_opts &o, long long t) {
    (void)t;
    const int *done = done_flag();
    Q.ensure((size_t)P * N);
    const double qbytes = 2.0 * P * N * sizeof(T);
    // MultUpdate's constructor floors both lambdas at sqrt(eps(T)) for obj = :div (src/multupd.jl:37-40); a raw C-ABI caller
    // gets the same floor here, so a zero column sum can never divide by zero
    const T lam_floor = std::sqrt(std::numeric_limits<T>::epsilon());
    const T lambda_h = std::max((T)o.lambda_h, lam_floor), lambda_w = std::max((T)o.lambda_w, lam_floor);
    const bool fused = !sharded() && div_fused;
    if (o.update_H) {
        const T *Wp = W[wcur].p;
        const T *Ho = H[hcur].p;
        T *Hn = H[hcur ^ 1].p;
        if (div_ieee) { EpiRatio<T, 0> er{X.p, Q.p, P, (T)o.delta}; gemm_wh("gemm_WH_ratio", Ho, Wp, er, done, qbytes); }    // :172-174
        else { EpiRatio<T, 1> er{X.p, Q.p, P, (T)o.delta}; gemm_wh("gemm_WH_ratio", Ho, Wp, er, done, qbytes); }
        if (fused) {
            // single GPU: the slab sum, the scaling, stop_condition's sums and sum(H, dims=2) for the W side in one pass
            wt_times(Wp, Q.p, false, done, /*keep_slabs=*/true);           // :175
            if (!div_sw_valid) {                                           // :176 (later iterations: from the W side's pass)
                timed("colsum_W", 0.0, (double)P * K * sizeof(T), [&] {
                    hipLaunchKernelGGL(col_sum_kernel<T>, dim3(stat_chunks_w, (unsigned)K), dim3(256), 0, stream, Wp, P, P, (int)K, stat_part.p, done);
                    hipLaunchKernelGGL(finalize_partials_kernel<T>, dim3((unsigned)((K + 3) / 4)), dim3(256), 0, stream, stat_part.p, stat_chunks_w, (int)K,
                                       (int)K, svec.p, done);
                });
            }
            timed("div_update_H", 0.0, (3.0 + h_num_nslab()) * K * N * sizeof(T), [&] {   // :177-179
                hipLaunchKernelGGL(div_h_fused_kernel<T>, dim3(stat_chunks_h), dim3(256), 0, stream, Hn, Ho, h_num(), h_num_nslab(), h_stride, svec.p, k, n,
                                   N, K, (int)K, lambda_h, stat_part.p, done);
                hipLaunchKernelGGL(finalize_div_kernel<T>, dim3((unsigned)((3 * K + 3) / 4)), dim3(256), 0, stream, stat_part.p, stat_chunks_h, (int)K, hstat.p,
                                   sH_p, done);
                HIP_TRY(hipGetLastError());
            });
            div_sh_valid = true;
            hcur ^= 1;
        } else {
        wt_times(Wp, Q.p, false, done);                                    // :175
        timed("colsum_W", 0.0, (double)P * K * sizeof(T), [&] {            // :176
            hipLaunchKernelGGL(col_sum_kernel<T>, dim3(stat_chunks_w, (unsigned)K), dim3(256), 0, stream, Wp, P, P, (int)K,
                               stat_part.p, done);
            hipLaunchKernelGGL(finalize_partials_kernel<T>, dim3((unsigned)((K + 3) / 4)), dim3(256), 0, stream, stat_part.p,
                               stat_chunks_w, (int)K, (int)K, svec.p, done);
        });
        timed("div_update_H", 0.0, 3.0 * K * N * sizeof(T), [&] {           // :177-179
            hipLaunchKernelGGL(div_update_kernel<T>, dim3(flat_grid(k * n)), dim3(256), 0, stream, Hn,
                               Ho, numH_p, svec.p, k, n, K, lambda_h, 1, done);
        });
        stats_h(Hn, Ho, done);
        hcur ^= 1;
        }
    }
    const T *Hp = H[hcur].p;
    const T *Wo = W[wcur].p;
    T *Wn = W[wcur ^ 1].p;
    if (div_ieee) { EpiRatio<T, 0> er{X.p, Q.p, P, (T)o.delta}; gemm_wh("gemm_WH_ratio", Hp, Wo, er, done, qbytes); }       // :184-186
    else { EpiRatio<T, 1> er{X.p, Q.p, P, (T)o.delta}; gemm_wh("gemm_WH_ratio", Hp, Wo, er, done, qbytes); }
    const bool rs = row_sharded();
    if (fused) {
        times_ht(Q.p, Hp, false, done, /*keep_slabs=*/true);               // :187
        if (!div_sh_valid) {                                               // :188 (update_H = false: H never changes)
            timed("rowsum_H", 0.0, (double)K * N * sizeof(T), [&] {
                hipLaunchKernelGGL(row_sum_kernel<T>, dim3(stat_chunks_h), dim3(256), 0, stream, Hp, N, K, (int)K, stat_part.p, done);
                hipLaunchKernelGGL(finalize_partials_kernel<T>, dim3((unsigned)((K + 3) / 4)), dim3(256), 0, stream, stat_part.p, stat_chunks_h, (int)K, (int)K,
                                   sH_p, done);
            });
            div_sh_valid = true;
        }
        timed("div_update_W", 0.0, (3.0 + w_num_nslab()) * P * K * sizeof(T), [&] {   // :189-191
            hipLaunchKernelGGL(div_w_fused_kernel<T>, dim3(stat_chunks_w, (unsigned)K), dim3(256), 0, stream, Wn, Wo, w_num(), w_num_nslab(), w_stride, sH_p, p,
                               k, P, P, (int)K, lambda_w, stat_part.p, done);
            hipLaunchKernelGGL(finalize_div_kernel<T>, dim3((unsigned)((3 * K + 3) / 4)), dim3(256), 0, stream, stat_part.p, stat_chunks_w, (int)K, wstat.p,
                               svec.p, done);
            HIP_TRY(hipGetLastError());
        });
        div_sw_valid = true;
        wcur ^= 1;
        return;
    }
    w_blocked = rs;
    times_ht(Q.p, Hp, false, done);                                        // :187
    w_blocked = false;
    T *sH = sH_p;                                                          // tail of the packed buffer
    timed("rowsum_H", 0.0, (double)K * N * sizeof(T), [&] {                // :188
        hipLaunchKernelGGL(row_sum_kernel<T>, dim3(stat_chunks_h), dim3(256), 0, stream, Hp, N, K, (int)K, stat_part.p, done);
        hipLaunchKernelGGL(finalize_partials_kernel<T>, dim3((unsigned)((K + 3) / 4)), dim3(256), 0, stream, stat_part.p,
                           stat_chunks_h, (int)K, (int)K, sH, done);
    });
    if (rs) {   // row-sharded W side: this rank scales its own rows only
        scatter_w_numerator(o.update_H != 0, done);
        const int64_t rows = std::max<int64_t>(0, std::min<int64_t>(Pc, p - row0));
        timed("div_update_W", 0.0, 3.0 * Pc * K * sizeof(T), [&] {           // :189-191
            if (rows > 0)
                hipLaunchKernelGGL(div_update_kernel<T>, dim3(flat_grid(rows * k)), dim3(256), 0, stream, Wn + row0, Wo + row0,
                                   numW_p + row0, sH, rows, k, P, lambda_w, 0, done);
        });
        HIP_TRY(hipGetLastError());
        stats_w_rows(Wn, Wo, done);
        gather_w_rows(Wn, true, done);
        wcur ^= 1;
        return;
    }
    allreduce_w_side(o.update_H != 0, done);
    timed("div_update_W", 0.0, 3.0 * P * K * sizeof(T), [&] {               // :189-191
        hipLaunchKernelGGL(div_update_kernel<T>, dim3(flat_grid(p * k)), dim3(256), 0, stream, Wn, Wo,
                           numW_p, sH, p, k, P, lambda_w, 0, done);
    });
    HIP_TRY(hipGetLastError());
    stats_w(Wn, Wo, done);
    wcur ^= 1;
}

// ---------------------------------------------------------------------------
// nmf_skeleton! (src/common.jl:45-89)
// ---------------------------------------------------------------------------
template <typename T> void Solver<T>::iterate(int alg, const nmfx_opts &o, nmfx_result *out, double *trace) {
    require_ready();
    if (o.maxiter < 1) throw StatusError{NMFX_ERR_BAD_ARG, "maxiter must be >= 1"};
    if (!(o.tol > 0)) throw StatusError{NMFX_ERR_BAD_ARG, "tol must be positive."};
    if (alg == NMFX_ALG_MULTMSE || alg == NMFX_ALG_MULTDIV) {
        if (!(o.lambda_w >= 0)) throw StatusError{NMFX_ERR_BAD_ARG, "lambda_w must be non-negative."};
        if (!(o.lambda_h >= 0)) throw StatusError{NMFX_ERR_BAD_ARG, "lambda_h must be non-negative."};
    }
    if (alg == NMFX_ALG_GREEDYCD) {   // src/greedycd.jl:27-28
        if (!(o.lambda_w >= 0)) throw StatusError{NMFX_ERR_BAD_ARG, "lambda_w must be non-negative."};
        if (!(o.lambda_h >= 0)) throw StatusError{NMFX_ERR_BAD_ARG, "lambda_h must be non-negative."};
    }
    if (alg < 0 || alg > NMFX_ALG_GREEDYCD) throw StatusError{NMFX_ERR_BAD_ARG, "Invalid algorithm."};
    if (o.precision != NMFX_PREC_FP32 && o.precision != NMFX_PREC_BF16X3) throw StatusError{NMFX_ERR_BAD_ARG, "Invalid value for precision."};
    if (o.pg_refresh < 0) throw StatusError{NMFX_ERR_BAD_ARG, "pg_refresh must be non-negative."};
    if (o.stop_sums != 0 && o.stop_sums != 1) throw StatusError{NMFX_ERR_BAD_ARG, "Invalid value for stop_sums."};
    if (o.stop_sums != 0 && sharded()) throw StatusError{NMFX_ERR_UNSUPPORTED, "stop_sums = 1 (sequential sums) is a one-GPU option"};
    if (o.h_solve < NMFX_HSOLVE_AUTO || o.h_solve > NMFX_HSOLVE_POTRS) throw StatusError{NMFX_ERR_BAD_ARG, "Invalid value for h_solve."};
    precision = o.precision;
    pipe_pending = false;
    defer_pending = false;
    check_fused = false;
    gramw_sharded_valid = false;
    w_res_blocked = w_std_stale = false;
    wb = 0;
    smallk_grams_valid = false;
    div_sw_valid = div_sh_valid = false;
    rsvd_ready = 0;   // the iteration overwrites the buffers a pending rsvd keeps its Q / B in
    HIP_TRY(hipSetDevice(device));
    std::memset(out, 0, sizeof *out);
    if (alg == NMFX_ALG_ALSPGRAD) { run_alspgrad(o, out, trace); return; }

    // host polls of the device stop flag: every `check_every` iterations, or (check_every <= 0) adaptively -- a window of 4
    // iterations that doubles (up to 256) while a window takes less than a millisecond of wall time: a poll is a host round trip,
    // which at small shapes (0.09 ms per iteration at 4096 x 4096, k = 64) is 5 % of the loop at a fixed 4.  Results do not depend
    // on it: iterations enqueued past the stop are no-ops.
    const bool adaptive = o.check_every <= 0;
    int window = adaptive ? 4 : o.check_every;
    long long next_poll = window;
    auto last_poll = std::chrono::steady_clock::now();
    const bool track = o.track_objective != 0;
    Ctrl init;
    std::memset(&init, 0, sizeof init);
    init.tolg = o.tolg;
    HIP_TRY(hipMemcpyAsync(ctrl, &init, sizeof init, hipMemcpyHostToDevice, stream));
    if (track) {
        trace_dev.ensure((size_t)o.maxiter + 1);
        std::vector<double> nanv((size_t)o.maxiter + 1, std::nan(""));
        HIP_TRY(hipMemcpyAsync(trace_dev.p, nanv.data(), nanv.size() * sizeof(double), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    // One block per CU instead of two for the big products of MultUpdate-MSE and CoordinateDescent, where the output is
    // exactly one 128 x 128 tile per CU (the headline shape): the 2-way split exists to put two blocks on a CU, but one block's waves
    // keep the matrix pipe as busy as two blocks' do -- a ProjectedALS product of this shape runs 892-895 us unsplit on 256 blocks
    // against 922-931 split, round 6 -- and there is one slab to write instead of two.  Measured (profiles/r06_big_products_unsplit_ab.jsonl):
    // MultUpdate-MSE 1.9384-1.9481 -> 1.9292-1.9344 ms per iteration, CoordinateDescent 2.069-2.073 -> 2.051-2.052; MultUpdate-Div
    // 4.046-4.053 -> 4.072-4.075 (SLOWER: left split); ProjectedALS keeps the short 2-per-CU grid its factorisations live under; GreedyCD
    // keeps the split too (its per-iteration time is a function of the greedy trajectory, which moves with the last bits of the
    // numerators: no like-for-like timing exists, and the step counts of earlier rounds' lines stay comparable).
    struct SplitGuard {
        Solver<T> &s; int sh, sw;
        ~SplitGuard() { s.s_h = sh; s.s_w = sw; s.skip_tree_stats = false; }
    } split_guard{*this, s_h, s_w};
    skip_tree_stats = o.stop_sums != 0;   // (one GPU only: checked above)
    if (unsplit_enabled && !sharded() && !use_bf16x3() && sizeof(T) == 4 && K % 128 == 0 &&
        (alg == NMFX_ALG_MULTMSE || alg == NMFX_ALG_CD)) {
        if ((N / 128) * (K / 128) == num_cu && s_h == 2) s_h = 1;
        if ((P / 128) * (K / 128) == num_cu && s_w == 2) s_w = 1;
    }
    // MultUpdate-MSE, general path, Float32: the X*H' product on the transposed images (solver.hpp: Xt)
    ht_active = false;
    if (alg == NMFX_ALG_MULTMSE && !smallk_ok() && !pipelined() && want_xt()) {
        ensure_xt();
        if (xt_valid) ht_active = refresh_ht();
    }
    if (alg == NMFX_ALG_CD && o.cd_shuffle != 0) prepare_cd_permutations(o);
    if (alg == NMFX_ALG_PROJALS) {
        // the buffer of the triangular inverse is zeroed ONCE per solve, here on the main stream: the register-resident factorisation
        // writes only the upper triangles of its diagonal blocks and trtri only the blocks above them, so the rest stays zero from one
        // iteration to the next (round 6: the per-iteration hipMemsetAsync in front of the factorisation is a 64-block launch, and on the
        // factorisation stream it waited for block slots under the product -- 363 us on average, in front of a 75-340 us potrf)
        work[1].ensure((size_t)K * K);
        HIP_TRY(hipMemsetAsync(work[1].p, 0, (size_t)K * K * sizeof(T), stream));
    }
    const int w0 = wcur, h0 = hcur;
    begin_iter_trace(o);
    HIP_TRY(hipEventRecord(ev_beg, stream));
    if (track) enqueue_objective(alg, o, trace_dev.p, done_flag());        // common.jl:56
    long long t = 0;
    while (t < o.maxiter) {
        ++t;
        switch (alg) {
            case NMFX_ALG_MULTMSE:
                if (pipelined()) enqueue_multmse_pipelined(o, t);
                else enqueue_multmse(o, t);
                break;
            case NMFX_ALG_MULTDIV: enqueue_multdiv(o, t); break;
            case NMFX_ALG_PROJALS: enqueue_projals(o, t); break;
            case NMFX_ALG_CD: enqueue_cd(o, t); break;
            case NMFX_ALG_GREEDYCD: enqueue_greedycd(o, t); break;
        }
        // pipelined exchange: the iteration's W is still travelling; its stop check runs when the next iteration has consumed
        // it.  Anything that needs the complete W (objective tracking, the host's poll) flushes the pipeline first.
        const bool poll = (t == next_poll) || t == o.maxiter;
        if (pipe_pending && (track || poll)) pipe_flush(o);
        if (defer_pending && (track || poll)) defer_flush();   // the deferred stop rule of this iteration, before the host looks at the flag
        // common.jl:79 -- enqueued BEFORE the stop check: the check raises the `done` flag that turns every later kernel
        // into a no-op, and the objective of the converging iteration itself must still be evaluated
        if (track) { w_sync(done_flag()); enqueue_objective(alg, o, trace_dev.p + t, done_flag()); }
        if (!pipe_pending) enqueue_check(o, t);                            // common.jl:73
        if (poll) {
            HIP_TRY(hipMemcpyAsync(ctrl_host, ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            if (ctrl_host->done || ctrl_host->status != 0) break;
            if (adaptive) {
                // (with a communicator every rank must poll -- and stop enqueueing -- at the SAME iterations, or the collectives of
                // the no-op iterations behind the stop no longer pair up: there the window grows by iteration count alone)
                const auto now = std::chrono::steady_clock::now();
                // (capped at 8: the collectives of the no-op iterations enqueued behind the stop still execute -- they must, every rank
                // has to issue the same sequence -- and in the fused row-sharded step they keep sum-all-reducing the stale Gram / statistics
                // buffers, multiplying them by nranks per no-op iteration; nothing reads those buffers after the stop, and with at most 7
                // such iterations they cannot overflow either: 16^7 x a Gram entry is far inside Float32)
                if (sharded()) { if (window < 8) window *= 2; }
                else if (std::chrono::duration<double>(now - last_poll).count() < 1e-3 && window < 256) window *= 2;
                last_poll = now;
            }
            next_poll = t + window;
        }
    }
    HIP_TRY(hipMemcpyAsync(ctrl_host, ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
#ifdef NMFX_POTRF_TIMING_LIB
    if (alg == NMFX_ALG_PROJALS && sizeof(T) == 4 && std::getenv("NMFX_POTRF_DUMP") != nullptr) {   // instrumented build: phases of the last potrf, wave by wave
        if (fstream != nullptr) HIP_TRY(hipStreamSynchronize(fstream));
        static long long dbg[2048];
        HIP_TRY(hipMemcpyFromSymbol(dbg, HIP_SYMBOL(nmfx_potrf_dbg), sizeof dbg));
        const long long t0 = dbg[256];
        std::fprintf(stderr, "  entry (cycles before the first step's stamp), waves 0..7: %lld %lld %lld %lld %lld %lld %lld %lld\n", t0 - dbg[128], t0 - dbg[129], t0 - dbg[130], t0 - dbg[131],
                     t0 - dbg[132], t0 - dbg[133], t0 - dbg[134], t0 - dbg[135]);
        for (int b = 0; b <= 8; ++b)
            for (int w = 0; w < 8; w += 7) {
                const long long *d = dbg + 256 + (b * 8 + w) * 8;
                std::fprintf(stderr, "  step %d wave %d: at %lld  A %lld  wait %lld  B %lld  wait %lld  C %lld (copy %lld)\n", b, w, d[0] - t0, d[1] - d[0], d[2] - d[1], d[3] - d[2],
                             d[4] - d[3], d[5] - d[4], w == 7 ? d[6] - d[4] : 0);
            }
    }
#endif
    const long long niters = ctrl_host->niters;
    // iterations enqueued after the stop were no-ops: the live buffers are those of iteration `niters`
    wcur = (int)((w0 + niters) & 1);
    hcur = o.update_H ? (int)((h0 + niters) & 1) : h0;
    if (w_res_blocked) {
        // W of iteration `niters` lives in the blocked buffer that iteration wrote (the in-place all-gathers of the no-op iterations
        // behind a stop re-deliver what is already there): unpack it into the standard layout for everything that follows
        wb = (int)(niters & 1);
        w_res_blocked = niters >= 1;
        w_std_stale = w_res_blocked;
        w_sync(nullptr);
        w_res_blocked = false;
    }
    if (ctrl_host->status == 0) {
        if (!track && final_objective) enqueue_objective(alg, o, obj_final.p, nullptr);       // common.jl:85-87
    }
    HIP_TRY(hipEventRecord(ev_end, stream));
    double objv = std::nan("");
    if (ctrl_host->status == 0) {
        if (track) {
            HIP_TRY(hipMemcpyAsync(&objv, trace_dev.p + niters, sizeof(double), hipMemcpyDeviceToHost, stream));
            if (trace) HIP_TRY(hipMemcpyAsync(trace, trace_dev.p, ((size_t)o.maxiter + 1) * sizeof(double), hipMemcpyDeviceToHost, stream));
        } else if (final_objective) {
            HIP_TRY(hipMemcpyAsync(&objv, obj_final.p, sizeof(double), hipMemcpyDeviceToHost, stream));
        }
    }
    HIP_TRY(hipStreamSynchronize(stream));
    if (comm) comm->health();   // a device-side wait of the peer exchange that timed out surfaces here
    end_iter_trace(o, niters);
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ev_beg, ev_end));
    out->niters = niters;
    out->converged = ctrl_host->converged;
    out->status = ctrl_host->status;
    out->objvalue = objv;
    out->seconds_loop = ms * 1e-3;
    out->inner_iters = ctrl_host->inner_iters;
    out->backtracks = ctrl_host->backtracks;
    out->final_tolg = ctrl_host->tolg;
    if (ctrl_host->status == NMFX_ERR_NOT_POSDEF) throw StatusError{NMFX_ERR_NOT_POSDEF, "matrix is not positive definite (potrf)"};
}

}  // namespace nmfx
