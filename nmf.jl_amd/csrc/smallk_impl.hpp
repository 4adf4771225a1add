// smallk_impl.hpp -- host side of the k <= 64 MultUpdate-MSE path (kernels and rationale: smallk.hpp).
#pragma once
#include "smallk.hpp"
#include "solver.hpp"

namespace nmfx {

template <typename T> void Solver<T>::enqueue_multmse_smallk(const nmfx_opts &o, long long t) {
    if constexpr (sizeof(T) == 4) {
        const int *done = done_flag();
        const int64_t stripes_h = N / 16, stripes_w = P / 16;
        smallk_slabs.ensure((size_t)std::max(stripes_h, stripes_w) * 4096);
        if (!smallk_attr_set) {      // per context: the attribute belongs to the device the context lives on
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&smallk_h_kernel<SMALLK_ST>), hipFuncAttributeMaxDynamicSharedMemorySize, SMALLK_H_LDS * 4));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&smallk_w_kernel<SMALLK_ST>), hipFuncAttributeMaxDynamicSharedMemorySize, SMALLK_W_LDS * 4));
            smallk_attr_set = true;
        }
        if (!smallk_grams_valid) {
            // first iteration of a solve: the Grams of the factors as they were handed in (later ones come out of the kernels)
            gram_w_only(W[wcur].p, done);
            if (!o.update_H) gram_h_only(H[hcur].p, done);
            smallk_grams_valid = true;
        }
        if (o.update_H) {
            const T *Ho = H[hcur].p;
            T *Hn = H[hcur ^ 1].p;
            timed("smallk_H", 2.0 * P * N * K + 2.0 * K * K * N * 2, (double)(P * N + P * K + 2 * K * N) * sizeof(T), [&] {
                hipLaunchKernelGGL(smallk_h_kernel<SMALLK_ST>, dim3((unsigned)stripes_h), dim3(SMALLK_THREADS), SMALLK_H_LDS * 4, stream, X.p, P, P, W[wcur].p, gramW_p,
                                   Ho, Hn, (float)o.lambda_h, (float)o.delta, smallk_slabs.p, stat_part.p, done);
                HIP_TRY(hipGetLastError());
            });
            timed("smallk_finish_H", 0.0, (double)stripes_h * (4096 * sizeof(T) + 128 * sizeof(double)), [&] {
                hipLaunchKernelGGL(smallk_finish_kernel, dim3(SMALLK_GRAM_BLOCKS + 32), dim3(256), 0, stream, gramH_p, smallk_slabs.p, (int)stripes_h, stat_part.p, hstat.p, done);
                HIP_TRY(hipGetLastError());
            });
            hcur ^= 1;
        }
        const T *Wo = W[wcur].p;
        T *Wn = W[wcur ^ 1].p;
        timed("smallk_W", 2.0 * P * N * K + 2.0 * K * K * P * 2, (double)(P * N + 2 * P * K) * sizeof(T), [&] {
            hipLaunchKernelGGL(smallk_w_kernel<SMALLK_ST>, dim3((unsigned)stripes_w), dim3(SMALLK_THREADS), SMALLK_W_LDS * 4, stream, X.p, P, N, H[hcur].p, gramH_p, Wo, Wn,
                               (float)o.lambda_w, (float)o.delta, smallk_slabs.p, stat_part.p, done);
            HIP_TRY(hipGetLastError());
        });
        // the stop rule of this iteration rides in the W-side finish launch unless the caller tracks the objective (check_kernel then
        // also records the verbose table's relchange column)
        const bool fuse_check = o.track_objective == 0 && o.stop_sums == 0;
        if (fuse_check && !smallk_ticket.p) smallk_ticket.ensure(1);
        timed("smallk_finish_W", 0.0, (double)stripes_w * (4096 * sizeof(T) + 128 * sizeof(double)), [&] {
            if (fuse_check)
                hipLaunchKernelGGL(smallk_finish_kernel, dim3(SMALLK_GRAM_BLOCKS + 32), dim3(256), 0, stream, gramW_p, smallk_slabs.p, (int)stripes_w, stat_part.p, wstat.p, done, ctrl,
                                   o.update_H ? hstat.p : (const double *)nullptr, (int)k, (float)o.tol, t, smallk_ticket.p);
            else
                hipLaunchKernelGGL(smallk_finish_kernel, dim3(SMALLK_GRAM_BLOCKS + 32), dim3(256), 0, stream, gramW_p, smallk_slabs.p, (int)stripes_w, stat_part.p, wstat.p, done);
            HIP_TRY(hipGetLastError());
        });
        check_fused = fuse_check;
        wcur ^= 1;
    }
}

}  // namespace nmfx
