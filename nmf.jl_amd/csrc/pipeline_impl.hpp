// pipeline_impl.hpp -- MultUpdate-MSE with the multi-GPU exchange pipelined against the two big products
// (NMFX_COMM_PIPELINED).  The reference has no distributed path: no reference counterpart.
//
// The row-sharded W side (solver_impl.hpp) is a dependency chain  X_g H_g' -> reduce-scatter -> W update -> all-gather -> W'X:
// nothing overlaps unless the chain is cut into pieces.  Here the P rows are cut into PIPE_C super-chunks; inside super-chunk c
// rank g owns the Pcc = P / (PIPE_C * G) rows [c*Rc + g*Pcc, ...).  Two streams: `stream` (kernels) and `cstream` (collectives).
//
//   W side, per super-chunk c:   X_g H_g' for the chunk's rows (its own launch, split-K slabs)  ->  blocked combine  ->  event
//                                cstream: reduce-scatter of the chunk           (runs under the launch of chunk c + 1)
//   then, per super-chunk:       rows of the update GEMM, statistics partials, pack  ->  event
//                                cstream: all-gather of the chunk               (runs under what follows on `stream`)
//   H side of the NEXT iteration, per super-chunk c:  wait for all-gather c, unpack its rows into W, the W'X split-K part whose
//                                contraction runs over exactly those rows (+ that part of W'W) -- chunk c + 1's all-gather
//                                runs under it.  Only then the stop check of the previous iteration (it needs the column
//                                statistics of ALL chunks), then the H update.
// A flush (objective tracking, the host's poll of the stop flag, the end of the solve) consumes whatever is in flight.
// Iterations behind a raised stop flag are no-ops like everywhere else (kernels check `done`; the collectives still run, on all
// ranks alike).
#pragma once
#include "solver.hpp"

namespace nmfx {

// rows chunk c of  numH += W[chunk,:]' * Bmat[chunk,:]  (+ the chunk's part of W'W as tail pieces): slabs [c*s, (c+1)*s)
template <typename T> void Solver<T>::wt_times_chunk(const T *Wp, const T *Bmat, int c, const int *done) {
    const int tiles = (int)((N / 128) * (K / 128));
    const int tt = (int)((K / 128) * (K / 128));
    const int s = pick_splits(tiles, Rc);
    const int per = tail_piece(tiles * s, tt, (int)(Rc / BK));
    const int pieces = (int)((Rc / BK + per - 1) / per);
    h_nslab = s * PIPE_C; h_stride = (int64_t)K * N;
    pipe_gram_pieces = pieces;
    const int64_t off = (int64_t)c * Rc;
    EpiStore<T> e{slabs.p + (size_t)c * s * h_stride, K, h_stride, nullptr};
    e.C2 = slabs.p + gram_slab_off + (size_t)c * pieces * K * K; e.ld2 = K; e.stride2 = (int64_t)K * K; e.r_off = N; e.c_off = 0;
    Seg sg;
    sg.A2 = Wp + off; sg.lda2 = P; sg.r_split = N; sg.tail_tiles = (int)(K / 128);
    gemm<KCONTIG, KCONTIG>("gemm_WtX", Bmat + off, P, N, Wp + off, P, K, Rc, s, true, e, done,
                           (double)(Rc * N + 2 * Rc * K) * sizeof(T), sg, 2.0 * K * K * Rc);
}

// rows chunk c of  numW[chunk,:] = Amat[chunk,:] * H'  (chunk 0 also carries H H' as tail pieces)
template <typename T> void Solver<T>::times_ht_chunk(const T *Amat, const T *Hp, int c, const int *done) {
    const int tiles = (int)((K / 128) * (Rc / 128));
    const int tt = (int)((K / 128) * (K / 128));
    const int s = pick_splits(tiles, N);
    w_nslab = s; w_stride = (int64_t)P * K;
    const int64_t off = (int64_t)c * Rc;
    T *reg = slabs.p + slab_w_off;
    EpiStore<T> e{reg + off, P, w_stride, nullptr};
    Seg sg;
    double extra = 0.0;
    if (c == 0) {
        const int per = tail_piece(tiles * s, tt, (int)(N / BK));
        pipe_gram_pieces = (int)((N / BK + per - 1) / per);
        e.C2 = slabs.p + gram_slab_off; e.ld2 = K; e.stride2 = (int64_t)K * K; e.r_off = 0; e.c_off = Rc;
        sg.B2 = Hp; sg.ldb2 = K; sg.c_split = Rc; sg.tail_tiles = (int)(K / 128);
        extra = 2.0 * K * K * N;
    }
    gemm<KSTRIDED, KSTRIDED>("gemm_XHt", Hp, K, K, Amat + off, P, Rc, N, s, false, e, done, (double)(Rc * N + K * N) * sizeof(T), sg, extra);
    if (c == 0) reduce_slabs_from("reduce_HHt", gramH_p, slabs.p + gram_slab_off, (int64_t)K * K, pipe_gram_pieces, done);
    // split-K combine of the chunk's rows straight into the blocked send buffer: piece (c, g) at ((c*G + g) * Pcc * K)
    hipLaunchKernelGGL(reduce_slabs_blocked_kernel<T>, dim3((unsigned)((Rc * K + 255) / 256)), dim3(256), 0, stream, numW_p, reg, P, K, Pcc,
                       w_nslab, w_stride, off, Rc, done);
    HIP_TRY(hipGetLastError());
}

// Consume the W that is in flight: per super-chunk wait for its all-gather, unpack its rows (+ add its statistics), and, if
// asked, launch the part of W'X / W'W that contracts over those rows.  Ends with the stop check of the iteration that produced
// this W (unless the caller runs that check itself, see run_check).
template <typename T> void Solver<T>::pipe_consume_w(const nmfx_opts &o, bool launch_wtx, bool run_check) {
    const int *done = done_flag();
    T *Wfull = W[wcur].p;
    for (int c = 0; c < PIPE_C; ++c) {
        HIP_TRY(hipStreamWaitEvent(stream, ev_ag[c], 0));
        hipLaunchKernelGGL(gathered_to_full_kernel<T>, dim3(flat_grid(Rc * K)), dim3(256), 0, stream, Wfull,
                           ag_recv.p + (size_t)c * nranks * agc_bytes, nranks, agc_bytes, P, K, Pcc, (int64_t)c * Rc, (int)(2 * K), wstat.p,
                           c > 0 ? 1 : 0, done);
        HIP_TRY(hipGetLastError());
        if (launch_wtx) wt_times_chunk(Wfull, X.p, c, done);
    }
    // run_check = false: the flush comes from iterate() for the CURRENT iteration, which enqueues the objective of that
    // iteration first and the stop check behind it (the check raises `done`, which turns every later launch into a no-op:
    // checked here, a converging iteration would lose its own objective value)
    if (run_check) enqueue_check(o, pipe_t);
    pipe_pending = false;
}

template <typename T> void Solver<T>::pipe_flush(const nmfx_opts &o) {
    if (pipe_pending) pipe_consume_w(o, false, /*run_check=*/false);
}

template <typename T> void Solver<T>::enqueue_multmse_pipelined(const nmfx_opts &o, long long t) {
    const int *done = done_flag();
    const bool was_pending = pipe_pending;
    if (o.update_H) {
        const T *Ho = H[hcur].p;
        T *Hn = H[hcur ^ 1].p;
        // W'X and W'W by row super-chunks: behind the previous iteration's all-gathers when one is in flight
        if (was_pending) pipe_consume_w(o, true, true);
        else for (int c = 0; c < PIPE_C; ++c) wt_times_chunk(W[wcur].p, X.p, c, done);
        reduce_slabs_from("reduce_WtW", gramW_p, slabs.p + gram_slab_off, (int64_t)K * K, pipe_gram_pieces * PIPE_C, done);
        reduce_slabs_from("reduce_WtX", numH_p, slabs.p, h_stride, h_nslab, done);
        EpiMultUpdate<T, 1> e{numH_p, 1, h_stride, Ho, Hn, K, (T)o.lambda_h, (T)o.delta, stat_part.p, (int)K};
        gemm<KCONTIG, KCONTIG>("gemm_WtWH_updH", Ho, K, N, gramW_p, K, K, K, 1, true, e, done, 4.0 * K * N * sizeof(T));
        stats_h_finalize(last_tiles_r, done);
        hcur ^= 1;
    } else if (was_pending) {
        pipe_consume_w(o, false, true);
    }
    const T *Hp = H[hcur].p;
    const T *Wo = W[wcur].p;
    T *Wn = W[wcur ^ 1].p;
    const int ct = CT;
    // ---- numerator chunks; chunk c's reduce-scatter runs on cstream under chunk c + 1's launch
    for (int c = 0; c < PIPE_C; ++c) {
        times_ht_chunk(X.p, Hp, c, done);
        HIP_TRY(hipEventRecord(ev_red[c], stream));
        HIP_TRY(hipStreamWaitEvent(cstream, ev_red[c], 0));
        if (c == 0) {
            comm->group_start();
            comm->reduce_scatter(numW_p, rs_out.p, (size_t)Pcc * K, ct, cstream);
            comm->all_reduce(gramH_p, (size_t)K * K, ct, false, cstream);
            if (o.update_H) comm->all_reduce(hstat.p, (size_t)2 * K, CT_F64, false, cstream);
            comm->group_end();
        } else {
            comm->reduce_scatter(numW_p + (size_t)c * Rc * K, rs_out.p + (size_t)c * Pcc * K, (size_t)Pcc * K, ct, cstream);
        }
        HIP_TRY(hipEventRecord(ev_rs[c], cstream));
    }
    // ---- the rank's rows of the W update, per super-chunk; chunk c's all-gather runs on cstream under what follows.
    // (All reduce-scatters first: the summed rows are unblocked INTO the buffer whose other regions are still send buffers.)
    for (int c = 0; c < PIPE_C; ++c) HIP_TRY(hipStreamWaitEvent(stream, ev_rs[c], 0));
    for (int c = 0; c < PIPE_C; ++c) {
        const int64_t r0 = (int64_t)c * Rc + (int64_t)rank * Pcc;
        hipLaunchKernelGGL(piece_to_rows_kernel<T>, dim3(flat_grid(Pcc * K)), dim3(256), 0, stream, numW_p, rs_out.p + (size_t)c * Pcc * K, P, K,
                           Pcc, r0, done);
        EpiMultUpdate<T, 0> e{numW_p + r0, 1, 0, Wo + r0, Wn + r0, P, (T)o.lambda_w, (T)o.delta, nullptr, 0};   // multupd.jl:110-114
        gemm<KSTRIDED, KSTRIDED>("gemm_WHHt_updW", gramH_p, K, K, Wo + r0, P, Pcc, K, 1, false, e, done, 3.0 * Pcc * K * sizeof(T));
        unsigned char *chunk = ag_send.p + (size_t)c * agc_bytes;
        const int chunks = (int)std::max<int64_t>(1, std::min<int64_t>(64, Pcc / 1024));
        hipLaunchKernelGGL(col_stats_kernel<T>, dim3(chunks, (unsigned)K), dim3(256), 0, stream, Wn + r0, Wo + r0, Pcc, P, (int)K, stat_part.p,
                           done);
        hipLaunchKernelGGL(finalize_partials_kernel<double>, dim3((unsigned)((2 * K + 3) / 4)), dim3(256), 0, stream, stat_part.p, chunks,
                           (int)(2 * K), (int)(2 * K), reinterpret_cast<double *>(chunk + (size_t)Pcc * K * sizeof(T)), done);
        hipLaunchKernelGGL(rows_to_piece_kernel<T>, dim3(flat_grid(Pcc * K)), dim3(256), 0, stream, reinterpret_cast<T *>(chunk), Wn, P, K, Pcc,
                           r0, done);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ev_pack[c], stream));
        HIP_TRY(hipStreamWaitEvent(cstream, ev_pack[c], 0));
        comm->all_gather(chunk, ag_recv.p + (size_t)c * nranks * agc_bytes, agc_bytes, CT_BYTE, cstream);
        HIP_TRY(hipEventRecord(ev_ag[c], cstream));
    }
    wcur ^= 1;
    pipe_pending = true;
    pipe_t = t;
}

}  // namespace nmfx
