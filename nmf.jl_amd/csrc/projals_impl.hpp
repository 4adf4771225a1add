// projals_impl.hpp -- ProjectedALS kernel sequence.  update_wh!(::ProjectedALSUpd), src/projals.jl:76-107:
//   H <- max(0, (W'W + lh I)^-1 W'X)      pdsolve!  (potrf! + potrs!, src/utils.jl:63-70)
//   W <- max(0, XH' (HH' + lw I)^-1)      pdrsolve! (potrf! + potri! + copytri! + mul!, src/utils.jl:72-84)
// Device form: U = potrf(A) (LDS-blocked, MFMA trailing update); Uinv = trtri(U) (blocked by 32, MFMA);  the H solve is Uinv*(Uinv'*B) (two k x k x n MFMA GEMMs
// in place of the two triangular substitutions of potrs!), the W side forms inv(A) = Uinv*Uinv' exactly
// like potri! and multiplies (MFMA GEMM) like the reference's mul!.
#pragma once
#include "chol.hpp"
#include "solver.hpp"

namespace nmfx {

template <typename T> void Solver<T>::enqueue_projals(const nmfx_opts &o, long long t) {
    (void)t;
    const int *done = done_flag();
    const size_t kk = (size_t)K * K;
    work[0].ensure((size_t)K * N);   // Y = Uinv' * W'X
    work[1].ensure(kk);              // Uinv
    work[2].ensure(kk);              // inv(HH' + lw I)
    T *Y = work[0].p, *Uinv = work[1].p, *invA = work[2].p;
    // potrf: 32 x 32 diagonal block + 32 x kp row panel in LDS (kp = k rounded up to 32)
    const size_t kp32 = (size_t)(k + 31) / 32 * 32;
    const size_t lds32 = ((size_t)(32 * 32 + 32 * kp32) * sizeof(T) + 15) / 16 * 16 + 16;
    const size_t lds_tri = (size_t)((k + 31) / 32 + 4) * 1024 * sizeof(T);   // finished tiles of a block column + 4 partial tiles
    if (lds_tri > 160 * 1024) throw StatusError{NMFX_ERR_UNSUPPORTED, "projals: k too large for the blocked triangular inverse"};
    if (lds32 > 160 * 1024) throw StatusError{NMFX_ERR_UNSUPPORTED, "projals: k too large for the single-workgroup Cholesky panel (k <= 1248 f32 / 608 f64)"};
    auto factor = [&](T *A, T lambda, const char *tag_potrf, const char *tag_trtri) {
        timed(tag_potrf, (double)k * k * k / 3.0, 0.0, [&] {
            if (lambda != (T)0)   // adddiag! skips lambda == 0 (src/utils.jl:18)
                hipLaunchKernelGGL(adddiag_kernel<T>, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, stream, A, K, (int)k, lambda, done);
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&potrf_upper_kernel<T>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds32));
            hipLaunchKernelGGL((potrf_upper_kernel<T>), dim3(1), dim3(1024), lds32, stream, A, K, (int)k, ctrl, (int)NMFX_ERR_NOT_POSDEF);
            HIP_TRY(hipGetLastError());
        });
        timed(tag_trtri, (double)k * k * k / 3.0, 0.0, [&] {
            HIP_TRY(hipMemsetAsync(Uinv, 0, kk * sizeof(T), stream));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&trtri_offdiag_kernel<T>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tri));
            const unsigned nblk = (unsigned)((k + 31) / 32);
            hipLaunchKernelGGL((trtri_diag_kernel<T>), dim3(nblk), dim3(64), 0, stream, A, Uinv, K, (int)k, done);
            hipLaunchKernelGGL((trtri_offdiag_kernel<T>), dim3(nblk), dim3(256), lds_tri, stream, A, Uinv, K, (int)k, done);
            HIP_TRY(hipGetLastError());
        });
    };
    if (o.update_H) {
        const T *Wp = W[wcur].p;
        const T *Ho = H[hcur].p;
        T *Hn = H[hcur ^ 1].p;
        wt_times(Wp, X.p, true, done);                                     // :92 W'W, :93 H <- W'X (one launch)
        factor(gramW_p, (T)o.lambda_h, "potrf_WtW", "trtri_WtW");                 // :92 adddiag!, :94 potrf!
        {   // :94 potrs!:  Y = Uinv' B ;  H = max(0, Uinv Y)   (:95 projectnn!)
            EpiStore<T> e1{Y, K, 0, nullptr};
            gemm<KCONTIG, KCONTIG>("gemm_UinvtB", numH_p, K, N, Uinv, K, K, K, 1, true, e1, done, 2.0 * K * N * sizeof(T));
            EpiClampStore<T> e2{Hn, K};
            gemm<KCONTIG, KSTRIDED>("gemm_UinvY_clampH", Y, K, N, Uinv, K, K, K, 1, true, e2, done, 2.0 * K * N * sizeof(T));
        }
        stats_h(Hn, Ho, done);
        hcur ^= 1;
    }
    const T *Hp = H[hcur].p;
    const T *Wo = W[wcur].p;
    T *Wn = W[wcur ^ 1].p;
    const bool rs = row_sharded();
    w_blocked = rs;
    times_ht(X.p, Hp, true, done);                                         // :100 HH', :101 XH' (one launch)
    w_blocked = false;
    if (rs) scatter_w_numerator(o.update_H != 0, done);                    // sharded: numerator rows of this rank + summed HH'
    else allreduce_w_side(o.update_H != 0, done);
    factor(gramH_p, (T)o.lambda_w, "potrf_HHt", "trtri_HHt");                     // :100 adddiag!, :102 potrf!
    {   // :102 potri! + copytri!: inv = Uinv Uinv' ; then W = max(0, XHt * inv)   (:103 projectnn!)
        EpiStore<T> e1{invA, K, 0, nullptr};
        gemm<KSTRIDED, KSTRIDED>("gemm_potri", Uinv, K, K, Uinv, K, K, K, 1, true, e1, done, 2.0 * K * K * sizeof(T));
        if (rs) {   // rows of W are independent: this rank forms ITS Pc rows, the all-gather re-assembles W
            EpiClampStore<T> e2{Wn + row0, P};
            gemm<KSTRIDED, KSTRIDED>("gemm_XHtInv_clampW", invA, K, K, numW_p + row0, P, Pc, K, 1, false, e2, done, 2.0 * Pc * K * sizeof(T));
        } else {
            EpiClampStore<T> e2{Wn, P};
            gemm<KSTRIDED, KSTRIDED>("gemm_XHtInv_clampW", invA, K, K, numW_p, P, P, K, 1, false, e2, done, 2.0 * P * K * sizeof(T));
        }
    }
    if (rs) {
        stats_w_rows(Wn, Wo, done);
        gather_w_rows(Wn, true, done);
    } else {
        stats_w(Wn, Wo, done);
    }
    wcur ^= 1;
}

}  // namespace nmfx
