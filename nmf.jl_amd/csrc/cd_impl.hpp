// cd_impl.hpp -- kernel sequences of CoordinateDescent (src/coorddesc.jl:160-181) and GreedyCD (src/greedycd.jl:165-177).
// Both update W FIRST (from the current H), then H (from the new W) -- the opposite order of the other updaters.
//   W side:  [X;H] H'  -> Z = XH' (p x k), P = HH' (k x k)   one fused GEMM launch (+ the packed all-reduce when sharded)
//   H side:  [X;W]' W  -> Z = (W'X) viewed as X'W (n x k),   P = W'W
// then the row-parallel sweep of cd.hpp over the samples (rows of W / columns of H).
#pragma once
#include "cd.hpp"
#include "solver.hpp"

namespace nmfx {

template <typename T> template <typename F> void Solver<T>::with_kmax(F &&f) {
    if (k <= 64) f(std::integral_constant<int, 1>{});
    else if (k <= 128) f(std::integral_constant<int, 2>{});
    else if (k <= 256) f(std::integral_constant<int, 4>{});
    else if (k <= 512) f(std::integral_constant<int, 8>{});
    else if (k <= 1024) f(std::integral_constant<int, 16>{});
    else throw StatusError{NMFX_ERR_UNSUPPORTED, "internal: with_kmax beyond the register forms (k > 1024 runs the LDS forms)"};
}
// GreedyCD: also 3 slots (k in 129..192) -- its fast form needs every slot live (cd.hpp, FULL)
template <typename T> template <typename F> void Solver<T>::with_kmax_greedy(F &&f) {
    if (k > 128 && k <= 192) f(std::integral_constant<int, 3>{});
    else with_kmax(std::forward<F>(f));
}
// k > 1024 (or NMFX_CD_LDS=1): the LDS forms of the sweeps (cd.hpp), one wave per sample row
template <typename T> bool Solver<T>::cd_use_lds() const { return k > 1024 || cd_force_lds; }
static inline void cd_lds_check(size_t bytes) {
    if (bytes > 160 * 1024) throw StatusError{NMFX_ERR_UNSUPPORTED, "cd / greedycd: k too large (a sample row's component vectors no longer fit 160 KiB of LDS)"};
}

// ---------------------------------------------------------------------------
// CoordinateDescent
// ---------------------------------------------------------------------------
template <typename T>
void Solver<T>::cd_sweep(SampleView<const T> Zo, SampleView<T> Zn, SampleView<const T> Num, const T *Pm, int64_t nsamples, T l1,
                         const int *done) {
    if (cd_use_lds()) {
        const int kp = (int)((k + 63) / 64 * 64);
        const size_t lds = (size_t)2 * kp * sizeof(T);
        cd_lds_check(lds);
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&cd_sweep_lds_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (nsamples > 0)
            hipLaunchKernelGGL((cd_sweep_lds_kernel<T>), dim3((unsigned)nsamples), dim3(64), lds, stream, Zo, Zn, Num, Pm, K, nsamples, (int)k, l1, done);
        return;
    }
    {
        // the blocked sweep (matrix-core gradient per 16 coordinates, cd.hpp) wherever its tile of W fits LDS: k <= 512.  Its run time is
        // that of ONE workgroup's rows whatever the number of workgroups (<= one per CU), so it also wins on small problems
        // (measured in Float32, ms per iteration old -> new: 1024^2 k=64 0.131 -> 0.109; 4096^2 k=256 0.469 -> 0.342; 8192 x 2048 k=512
        // 1.755 -> 0.800; 16384^2 k=256 2.611 -> 2.237).  NMFX_CD_BLOCKED=0 keeps the row-chain kernels.
        const bool shape = (K == 64 || K == 128 || K == 192 || K == 256 || K == 320 || K == 384 || K == 512);
        if (shape && cd_blocked != 0) {
            auto launch = [&](auto KGC, auto ROWSC) {
                constexpr int KG = decltype(KGC)::value, ROWS = decltype(ROWSC)::value;
                const size_t lds = cd_blocked_lds_bytes<T>(K, ROWS);
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&cd_sweep_blocked_kernel<T, KG, ROWS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                if (nsamples > 0)
                    hipLaunchKernelGGL((cd_sweep_blocked_kernel<T, KG, ROWS>), dim3((unsigned)((nsamples + ROWS - 1) / ROWS)), dim3(ROWS * 4), lds, stream, Zo, Zn, Num, Pm,
                                       K, nsamples, (int)k, l1, done);
            };
            constexpr int GR = 64 / (int)sizeof(T);      // contraction indices per fragment group: 16 (Float32) / 8 (Float64)
            using R64 = std::integral_constant<int, 64>;
            using R32 = std::integral_constant<int, 32>;
            if (K == 64) launch(std::integral_constant<int, 64 / GR>{}, R64{});
            else if (K == 128) launch(std::integral_constant<int, 128 / GR>{}, R64{});
            else if (K == 192) launch(std::integral_constant<int, 192 / GR>{}, R64{});
            else if (K == 256) launch(std::integral_constant<int, 256 / GR>{}, R64{});
            else if constexpr (sizeof(T) == 4) {
                if (K == 320) launch(std::integral_constant<int, 320 / GR>{}, R64{});
                else if (K == 384) launch(std::integral_constant<int, 384 / GR>{}, R64{});
                else launch(std::integral_constant<int, 512 / GR>{}, R64{});
            } else {                                      // Float64: 64 rows x 320 (384, 512) components no longer fit 160 KiB
                if (K == 320) launch(std::integral_constant<int, 320 / GR>{}, R32{});
                else if (K == 384) launch(std::integral_constant<int, 384 / GR>{}, R32{});
                else launch(std::integral_constant<int, 512 / GR>{}, R32{});
            }
            return;
        }
    }
    const int kpl = (int)(K / 16);                       // K is 64 or a multiple of 128: kpl is a multiple of 4
    const int reg_budget = (sizeof(T) == 4) ? 32 : 16;   // components per lane kept in registers (w, z, a: 3 arrays)
    if (kpl <= reg_budget) {
        const unsigned blocks = (unsigned)((nsamples + 15) / 16);
        auto launch = [&](auto KP) {
            constexpr int KPLMAX = decltype(KP)::value;
            hipLaunchKernelGGL((cd_sweep16_kernel<T, KPLMAX>), dim3(blocks), dim3(256), 0, stream, Zo, Zn, Num, Pm, K, nsamples, (int)k, kpl,
                               l1, done);
        };
        if (kpl <= 4) launch(std::integral_constant<int, 4>{});
        else if (kpl <= 8) launch(std::integral_constant<int, 8>{});
        else if (kpl <= 16) launch(std::integral_constant<int, 16>{});
        else if constexpr (sizeof(T) == 4) launch(std::integral_constant<int, 32>{});   // (<double, 32> is never needed -- and never instantiated: it sends the register allocator into a minutes-long spin)
    } else {
        with_kmax([&](auto KM) {
            constexpr int KMAX = decltype(KM)::value, R = (KMAX <= 4) ? 4 : 2;
            const unsigned blocks = (unsigned)((nsamples + 4 * R - 1) / (4 * R));
            hipLaunchKernelGGL((cd_sweep_kernel<T, KMAX, R>), dim3(blocks), dim3(256), 0, stream, Zo, Zn, Num, Pm, K, nsamples, (int)k, l1, done);
        });
    }
}

// The component orders of a CoordinateDescent(shuffle = true) solve: call c = 2*(t-1) + side sweeps in the order that sorts the keys
// Philox4x32-10((i, c, 4, 0), key = cd_shuffle)[0], i < k (include/nmfx.h; NumPy twin: tests/philox_ref.py::cd_permutation).
// Generated lazily, a window of CD_PERM_WINDOW iterations at a time (2 * window * k ints on the device): a solve with a huge
// maxiter and a tolerance that stops it after a few iterations pays for one window, not for 2 * maxiter * k orders up front.
// The upload is stream-ordered behind the sweeps that still read the previous window.
template <typename T> void Solver<T>::prepare_cd_permutations(const nmfx_opts &o) {
    (void)o;
    cd_perm_w0 = -1;   // no window resident: the first iteration generates [0, CD_PERM_WINDOW)
}
template <typename T> const int *Solver<T>::cd_permutation_window(const nmfx_opts &o, long long t) {
    const long long it = t - 1;
    if (cd_perm_w0 < 0 || it < cd_perm_w0 || it >= cd_perm_w0 + CD_PERM_WINDOW) {
        const uint64_t key = (uint64_t)(int64_t)o.cd_shuffle;   // sign-extended 32-bit field
        const long long w0 = it, w1 = std::min<long long>((long long)o.maxiter, w0 + CD_PERM_WINDOW);
        const size_t calls = (size_t)2 * (size_t)(w1 - w0);
        cd_perm_host.resize(calls * (size_t)k);
        std::vector<std::pair<uint32_t, int>> keys((size_t)k);
        for (size_t cc = 0; cc < calls; ++cc) {
            const uint64_t c = (uint64_t)2 * (uint64_t)w0 + cc;
            if (c >> 32) throw StatusError{NMFX_ERR_UNSUPPORTED, "cd: shuffle = true supports at most 2^31 iterations (the call index is a 32-bit Philox counter word)"};
            for (int i = 0; i < (int)k; ++i) {
                uint32_t w[4];
                philox4x32_10((uint32_t)i, (uint32_t)c, 4u, 0u, (uint32_t)key, (uint32_t)(key >> 32), w);
                keys[(size_t)i] = {w[0], i};
            }
            std::sort(keys.begin(), keys.end());
            for (int i = 0; i < (int)k; ++i) cd_perm_host[cc * (size_t)k + (size_t)i] = keys[(size_t)i].second;
        }
        cd_perm.ensure((size_t)2 * CD_PERM_WINDOW * (size_t)k);
        HIP_TRY(hipMemcpyAsync(cd_perm.p, cd_perm_host.data(), cd_perm_host.size() * sizeof(int), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));   // pageable source: the copy must have left the host vector before it is refilled
        cd_perm_w0 = w0;
    }
    return cd_perm.p + (size_t)(2 * (it - cd_perm_w0)) * (size_t)k;
}

// cd_sweep in the component order `perm` (nullptr: 1..k): rename in, sweep, rename out (cd.hpp)
template <typename T>
void Solver<T>::cd_sweep_ordered(SampleView<const T> Zo, SampleView<T> Zn, SampleView<const T> Num, const T *Pm, int64_t nsamples, T l1,
                                 const int *perm, int64_t offset, const int *done) {
    if (perm == nullptr) { cd_sweep(Zo, Zn, Num, Pm, nsamples, l1, done); return; }
    const size_t big = (size_t)std::max(P * K, K * N);
    work[4].ensure(big);
    work[5].ensure(big);
    work[6].ensure((size_t)K * K);
    SampleView<T> Wp{work[4].p + offset, Zo.ss, Zo.cs}, Np{work[5].p + offset, Zo.ss, Zo.cs};
    const unsigned grid = flat_grid(nsamples * k);
    hipLaunchKernelGGL(permute_components_kernel<T>, dim3(grid), dim3(256), 0, stream, Wp, Zo, perm, nsamples, (int)k, 0, done);
    hipLaunchKernelGGL(permute_components_kernel<T>, dim3(grid), dim3(256), 0, stream, Np, Num, perm, nsamples, (int)k, 0, done);
    hipLaunchKernelGGL(permute_gram_kernel<T>, dim3(flat_grid(k * k)), dim3(256), 0, stream, work[6].p, Pm, K, perm, (int)k, done);
    cd_sweep(SampleView<const T>{Wp.p, Wp.ss, Wp.cs}, Wp, SampleView<const T>{Np.p, Np.ss, Np.cs}, work[6].p, nsamples, l1, done);
    hipLaunchKernelGGL(permute_components_kernel<T>, dim3(grid), dim3(256), 0, stream, Zn, SampleView<const T>{Wp.p, Wp.ss, Wp.cs}, perm,
                       nsamples, (int)k, 1, done);
    HIP_TRY(hipGetLastError());
}

template <typename T> void Solver<T>::enqueue_cd(const nmfx_opts &o, long long t) {
    const int *done = done_flag();
    const int *perm_w = o.cd_shuffle ? cd_permutation_window(o, t) : nullptr;
    const int *perm_h = o.cd_shuffle ? perm_w + (size_t)k : nullptr;
    {   // ---- W (coorddesc.jl:166): HHt = H*Ht, XHt = X*Ht (:109-115)
        const T *Hp = H[hcur].p;
        const T *Wo = W[wcur].p;
        T *Wn = W[wcur ^ 1].p;
        // multi-GPU: the rows of W do not interact in the sweep, so with the row-sharded W side this rank
        // sweeps only ITS rows [row0, row0 + rows): numerator by reduce-scatter, rows re-assembled by the all-gather
        const bool rs = row_sharded();
        const int64_t r0 = rs ? row0 : 0, rows = rs ? std::max<int64_t>(0, std::min<int64_t>(Pc, p - row0)) : p;
        w_blocked = rs;
        times_ht(X.p, Hp, true, done, false, ht_for(Hp, done));
        w_blocked = false;
        if (rs) scatter_w_numerator(false, done);
        else allreduce_w_side(false, done);
        if (o.l2_w > 0)   // :118-120
            hipLaunchKernelGGL(adddiag_kernel<T>, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, stream, gramH_p, K, (int)k, (T)o.l2_w, done);
        timed("cd_sweep_W", 2.0 * rows * k * k, 3.0 * rows * K * sizeof(T), [&] {
            if (rows > 0)
                cd_sweep_ordered(SampleView<const T>{Wo + r0, 1, P}, SampleView<T>{Wn + r0, 1, P}, SampleView<const T>{numW_p + r0, 1, P}, gramH_p,
                                 rows, (T)o.l1_w, perm_w, r0, done);
            HIP_TRY(hipGetLastError());
        });
        if (rs) {
            stats_w_rows(Wn, Wo, done);
            gather_w_rows(Wn, true, done);
        } else {
            stats_w(Wn, Wo, done);
        }
        wcur ^= 1;
    }
    if (o.update_H) {   // ---- H (:169-174) through the transposed views
        const T *Wp = W[wcur].p;
        const T *Ho = H[hcur].p;
        T *Hn = H[hcur ^ 1].p;
        wt_times(Wp, X.p, true, done);
        if (o.l2_h > 0)
            hipLaunchKernelGGL(adddiag_kernel<T>, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, stream, gramW_p, K, (int)k, (T)o.l2_h, done);
        timed("cd_sweep_H", 2.0 * n * k * k, 3.0 * K * N * sizeof(T), [&] {
            cd_sweep_ordered(SampleView<const T>{Ho, K, 1}, SampleView<T>{Hn, K, 1}, SampleView<const T>{numH_p, K, 1}, gramW_p, n, (T)o.l1_h,
                             perm_h, 0, done);
            HIP_TRY(hipGetLastError());
        });
        stats_h(Hn, Ho, done);
        allreduce_hstat(done);
        hcur ^= 1;
    }
}

// ---------------------------------------------------------------------------
// GreedyCD
// ---------------------------------------------------------------------------
template <typename T>
void Solver<T>::greedy_side(const char *tag, SampleView<const T> Zo, SampleView<T> Zn, SampleView<const T> G, const T *Pm, int64_t nsamples,
                            T lambda, bool sharded_samples, const int *done) {
    const T epsT = std::numeric_limits<T>::epsilon();
    const unsigned blocks = (unsigned)std::max<int64_t>(1, (nsamples + 3) / 4);   // >= 1: a rank without samples still takes part in the p_init all-reduce
    // (sized once for BOTH sides: growing it between the W side and the H side would hipFree -- a device-wide synchronisation -- in the
    // middle of an iteration, which deadlocks several in-process ranks whose peers spin on the device for this rank's next flag)
    work[3].ensure((size_t)std::max(P, N) + 16);
    greedy_queue.ensure((size_t)GREEDY_NQ * GREEDY_QSTRIDE);
    T *part = work[3].p, *pinit = work[3].p + blocks;
    if (cd_use_lds()) {
        const int kp = (int)((k + 63) / 64 * 64);
        const size_t lds = GreedyLds<T>::bytes(kp);
        cd_lds_check(lds);
        T *part1 = work[3].p, *pinit1 = work[3].p + std::max<int64_t>(1, nsamples);
        timed(tag, 0.0, 4.0 * (double)nsamples * K * sizeof(T), [&] {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&greedy_pinit_lds_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&greedy_sweep_lds_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (nsamples > 0)
                hipLaunchKernelGGL((greedy_pinit_lds_kernel<T>), dim3((unsigned)nsamples), dim3(64), lds, stream, Zo, G, Pm, K, nsamples, (int)k, lambda, epsT,
                                   part1, done);
            hipLaunchKernelGGL(greedy_pinit_reduce_kernel<T>, dim3(1), dim3(256), 0, stream, part1, (int)nsamples, pinit1, (int *)nullptr, done);
            if (sharded_samples && sharded()) comm->all_reduce(pinit1, 1, CT, true, stream);
            if (nsamples > 0)
                hipLaunchKernelGGL((greedy_sweep_lds_kernel<T>), dim3((unsigned)nsamples), dim3(64), lds, stream, Zo, Zn, G, Pm, K, nsamples, (int)k, lambda,
                                   epsT, pinit1, &ctrl->inner_iters, done);
            HIP_TRY(hipGetLastError());
        });
        return;
    }
    timed(tag, 0.0, 4.0 * (double)nsamples * K * sizeof(T), [&] {
        with_kmax_greedy([&](auto KM) {
            constexpr int KMAX = decltype(KM)::value;
            hipLaunchKernelGGL((greedy_pinit_kernel<T, KMAX>), dim3(blocks), dim3(256), 0, stream, Zo, G, Pm, K, nsamples, (int)k, lambda,
                               epsT, part, done);
            hipLaunchKernelGGL(greedy_pinit_reduce_kernel<T>, dim3(1), dim3(256), 0, stream, part, (int)blocks, pinit, greedy_queue.p, done);
            if (sharded_samples && sharded())   // p_init is the maximum over ALL samples (greedycd.jl:127-132)
                comm->all_reduce(pinit, 1, CT, true, stream);
            // persistent launch: the waves that are resident at once; the rows beyond them are handed out through greedy_queue (cd.hpp)
            auto launch = [&](auto full) {
                constexpr bool FULL = decltype(full)::value;
                static const int per_cu = [] {
                    int nb = 0;
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(&greedy_sweep_kernel<T, KMAX, FULL>), 256, 0) != hipSuccess || nb < 1) nb = 1;
                    return nb;
                }();
                // (NMFX_GREEDY_WGS_PER_CU=n, development switch: at most n resident workgroups per CU -- the occupancy probe of DESIGN.md
                // section 3.2: is the sweep bound by latency, i.e. by how many waves interleave on a SIMD, or by instruction issue?)
                // (the probe's 15-iteration run has its optimum at SIX workgroups per CU, 811 / 1520 us against 821 / 1591 for the W / H
                // sweeps; on the bench's standard run eight win again, 6.22-6.28 against 6.28-6.35 ms per iteration: the default stays all that fit)
                int cap = per_cu;
                if (const char *e = dev_env("NMFX_GREEDY_WGS_PER_CU")) cap = std::max(1, std::min(per_cu, std::atoi(e)));
                size_t pad = 0;
                if (cap < per_cu) {
                    pad = (size_t)(160 * 1024 / (cap + 1) + 1024) / 256 * 256;   // dynamic LDS that lets `cap` workgroups onto a CU, not cap + 1
                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&greedy_sweep_kernel<T, KMAX, FULL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad));
                }
                const unsigned sweep_blocks = (unsigned)std::min<int64_t>(blocks, (int64_t)cap * num_cu);
                hipLaunchKernelGGL((greedy_sweep_kernel<T, KMAX, FULL>), dim3(sweep_blocks), dim3(256), pad, stream, Zo, Zn, G, Pm, K, nsamples, (int)k,
                                   lambda, epsT, pinit, greedy_queue.p, &ctrl->inner_iters, done);
            };
            if ((k + 63) / 64 == KMAX) launch(std::true_type{});   // every slot live
            else launch(std::false_type{});
        });
        HIP_TRY(hipGetLastError());
    });
}

template <typename T> void Solver<T>::enqueue_greedycd(const nmfx_opts &o, long long t) {
    (void)t;
    const int *done = done_flag();
    {   // ---- W (greedycd.jl:168): P = Ht'Ht, Z = X*Ht, G = W*P - Z (:108-112)
        const T *Hp = H[hcur].p;
        const T *Wo = W[wcur].p;
        T *Wn = W[wcur ^ 1].p;
        // multi-GPU, row-sharded W side: this rank forms G and sweeps for ITS rows; p_init is the maximum over ALL rows
        // (greedycd.jl:127-132) -> one max all-reduce, like on the column-sharded H side
        const bool rs = row_sharded();
        const int64_t r0 = rs ? row0 : 0, rows = rs ? std::max<int64_t>(0, std::min<int64_t>(Pc, p - row0)) : p;
        w_blocked = rs;
        times_ht(X.p, Hp, true, done, false, ht_for(Hp, done));
        w_blocked = false;
        if (rs) scatter_w_numerator(false, done);
        else allreduce_w_side(false, done);
        work[0].ensure((size_t)std::max(P * K, K * N));
        T *G = work[0].p;
        EpiSubStore<T> e{numW_p + r0, G + r0, P};
        gemm<KSTRIDED, KSTRIDED>("gemm_WP_subZ", gramH_p, K, K, Wo + r0, P, rs ? Pc : P, K, 1, false, e, done, 3.0 * (rs ? Pc : P) * K * sizeof(T));
        greedy_side("greedy_W", SampleView<const T>{Wo + r0, 1, P}, SampleView<T>{Wn + r0, 1, P}, SampleView<const T>{G + r0, 1, P}, gramH_p, rows,
                    (T)o.lambda_w, rs, done);
        if (rs) {
            stats_w_rows(Wn, Wo, done);
            gather_w_rows(Wn, true, done);
        } else {
            stats_w(Wn, Wo, done);
        }
        wcur ^= 1;
    }
    if (o.update_H) {   // ---- H (:171-174)
        const T *Wp = W[wcur].p;
        const T *Ho = H[hcur].p;
        T *Hn = H[hcur ^ 1].p;
        wt_times(Wp, X.p, true, done);
        work[0].ensure((size_t)std::max(P * K, K * N));
        T *G = work[0].p;
        EpiSubStore<T> e{numH_p, G, K};
        gemm<KCONTIG, KCONTIG>("gemm_PH_subZ", Ho, K, N, gramW_p, K, K, K, 1, true, e, done, 3.0 * K * N * sizeof(T));
        greedy_side("greedy_H", SampleView<const T>{Ho, K, 1}, SampleView<T>{Hn, K, 1}, SampleView<const T>{G, K, 1}, gramW_p, n,
                    (T)o.lambda_h, true, done);
        stats_h(Hn, Ho, done);
        allreduce_hstat(done);
        hcur ^= 1;
    }
}

}  // namespace nmfx
