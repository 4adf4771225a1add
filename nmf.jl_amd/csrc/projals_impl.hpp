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

// U = potrf(A + lambda I) in place (upper triangle of A), Uinv = inv(U).  adddiag! (src/utils.jl:15-24) + potrf! of
// pdsolve! / pdrsolve! (src/utils.jl:63-84).  A non-positive pivot raises ctrl->status = NOT_POSDEF (PosDefException).
template <typename T> void Solver<T>::spd_factor(T *A, T lambda, T *Uinv, const char *tag_potrf, const char *tag_trtri, const int *done) {
    const size_t kk = (size_t)K * K;
    // potrf: 32 x 32 diagonal block + 32 x kp row panel in LDS (kp = k rounded up to 32)
    const size_t lds32 = potrf_lds_bytes();
    const size_t lds_tri = (size_t)((k + 31) / 32 + 4) * 1024 * sizeof(T);   // finished tiles of a block column + 4 partial tiles
    if (lds_tri > 160 * 1024) throw StatusError{NMFX_ERR_UNSUPPORTED, "projals: k too large for the blocked triangular inverse"};
    if (lds32 > 160 * 1024) throw StatusError{NMFX_ERR_UNSUPPORTED, "projals: k too large for the single-workgroup Cholesky panel (k <= 1248 f32 / 608 f64)"};
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
}

// pdsolve! (src/utils.jl:63-70) after the factorisation: out = inv(A) B for B, out K x N (ld K): Y = Uinv' B, out = Uinv Y
// (the two substitutions of potrs! as two k x k x n MFMA GEMMs); clamp = projectnn! fused into the store (src/utils.jl:34-41)
template <typename T> void Solver<T>::spd_solve_left(const T *Uinv, const T *B, T *Y, T *out, bool clamp, const int *done) {
    EpiStore<T> e1{Y, K, 0, nullptr};
    gemm<KCONTIG, KCONTIG>("gemm_UinvtB", B, K, N, Uinv, K, K, K, 1, true, e1, done, 2.0 * K * N * sizeof(T));
    if (clamp) {
        EpiClampStore<T> e2{out, K};
        gemm<KCONTIG, KSTRIDED>("gemm_UinvY_clampH", Y, K, N, Uinv, K, K, K, 1, true, e2, done, 2.0 * K * N * sizeof(T));
    } else {
        EpiStore<T> e2{out, K, 0, nullptr};
        gemm<KCONTIG, KSTRIDED>("gemm_UinvY", Y, K, N, Uinv, K, K, K, 1, true, e2, done, 2.0 * K * N * sizeof(T));
    }
}

// pdrsolve! (src/utils.jl:72-84) after the factorisation: inv = Uinv Uinv' (potri! + copytri!), out = A * inv for `rows` rows
// of A, out (ld P) -- mul! of the reference; clamp = projectnn!
template <typename T> void Solver<T>::spd_solve_right(const T *Uinv, T *invA, const T *A, T *out, int64_t rows, bool clamp, const int *done) {
    EpiStore<T> e1{invA, K, 0, nullptr};
    gemm<KSTRIDED, KSTRIDED>("gemm_potri", Uinv, K, K, Uinv, K, K, K, 1, true, e1, done, 2.0 * K * K * sizeof(T));
    if (clamp) {
        EpiClampStore<T> e2{out, P};
        gemm<KSTRIDED, KSTRIDED>("gemm_XHtInv_clampW", invA, K, K, A, P, rows, K, 1, false, e2, done, 2.0 * rows * K * sizeof(T));
    } else {
        EpiStore<T> e2{out, P, 0, nullptr};
        gemm<KSTRIDED, KSTRIDED>("gemm_XHtInv", invA, K, K, A, P, rows, K, 1, false, e2, done, 2.0 * rows * K * sizeof(T));
    }
}

// The SPD utilities of src/utils.jl through the C ABI (nmfx_pdsolve / nmfx_pdrsolve), on the kernels above.
template <typename T> void Solver<T>::pdsolve_host(int right, const void *A_host, const void *B_host, double lambda, void *X_host, bool clamp) {
    HIP_TRY(hipSetDevice(device));
    const size_t kk = (size_t)K * K;
    work[0].ensure((size_t)K * N);
    work[1].ensure(kk);
    work[2].ensure(kk);
    Ctrl init;
    std::memset(&init, 0, sizeof init);
    HIP_TRY(hipMemcpyAsync(ctrl, &init, sizeof init, hipMemcpyHostToDevice, stream));
    // the SPD matrix goes where projals keeps its Gram (gramW_p for the left solve, gramH_p for the right one), the other
    // operand where the numerator lives; everything zero-padded
    T *G = right ? gramH_p : gramW_p;
    HIP_TRY(hipMemsetAsync(G, 0, kk * sizeof(T), stream));
    HIP_TRY(hipMemcpy2DAsync(G, K * sizeof(T), right ? B_host : A_host, k * sizeof(T), k * sizeof(T), k, hipMemcpyHostToDevice, stream));
    spd_factor(G, (T)lambda, work[1].p, "potrf", "trtri", nullptr);
    have_F = false;   // the factor buffers are scratch for this call
    if (!right) {     // x <- inv(A) x, x is k x n
        HIP_TRY(hipMemsetAsync(numH_p, 0, (size_t)K * N * sizeof(T), stream));
        HIP_TRY(hipMemcpy2DAsync(numH_p, K * sizeof(T), B_host, k * sizeof(T), k * sizeof(T), n, hipMemcpyHostToDevice, stream));
        spd_solve_left(work[1].p, numH_p, work[0].p, H[0].p, clamp, nullptr);
        HIP_TRY(hipMemcpy2DAsync(X_host, k * sizeof(T), H[0].p, K * sizeof(T), k * sizeof(T), n, hipMemcpyDeviceToHost, stream));
    } else {          // x <- A inv(B), A and x are p x k
        HIP_TRY(hipMemsetAsync(numW_p, 0, (size_t)P * K * sizeof(T), stream));
        HIP_TRY(hipMemcpy2DAsync(numW_p, P * sizeof(T), A_host, p * sizeof(T), p * sizeof(T), k, hipMemcpyHostToDevice, stream));
        spd_solve_right(work[1].p, work[2].p, numW_p, W[0].p, P, clamp, nullptr);
        HIP_TRY(hipMemcpy2DAsync(X_host, p * sizeof(T), W[0].p, P * sizeof(T), p * sizeof(T), k, hipMemcpyDeviceToHost, stream));
    }
    HIP_TRY(hipMemcpyAsync(ctrl_host, ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    // the padded parts of W[0] / H[0] were written by the GEMM epilogues (zeros times zeros): still zero
    if (ctrl_host->status == NMFX_ERR_NOT_POSDEF) throw StatusError{NMFX_ERR_NOT_POSDEF, "matrix is not positive definite (potrf)"};
}

template <typename T> void Solver<T>::enqueue_projals(const nmfx_opts &o, long long t) {
    (void)t;
    const int *done = done_flag();
    const size_t kk = (size_t)K * K;
    work[0].ensure((size_t)K * N);   // Y = Uinv' * W'X
    work[1].ensure(kk);              // Uinv
    work[2].ensure(kk);              // inv(HH' + lw I)
    T *Y = work[0].p, *Uinv = work[1].p, *invA = work[2].p;
    // (Running the factorisation on a second stream UNDER the big GEMM was built and measured: it never becomes co-resident --
    // two 184-register GEMM waves per SIMD are allocated as 2 x 256 and fill the register file, so the Cholesky workgroup only
    // starts when GEMM blocks drain; 2.76 ms per iteration either way.  DESIGN.md section 3.2.)
    const bool rs = row_sharded();
    if (o.update_H) {
        const T *Wp = W[wcur].p;
        const T *Ho = H[hcur].p;
        T *Hn = H[hcur ^ 1].p;
        wt_times(Wp, X.p, true, done);                                     // :92 W'W, :93 H <- W'X (one launch)
        spd_factor(gramW_p, (T)o.lambda_h, Uinv, "potrf_WtW", "trtri_WtW", done);   // :92 adddiag!, :94 potrf!
        spd_solve_left(Uinv, numH_p, Y, Hn, true, done);                   // :94 potrs!, :95 projectnn!
        stats_h(Hn, Ho, done);
        hcur ^= 1;
    }
    const T *Hp = H[hcur].p;
    const T *Wo = W[wcur].p;
    T *Wn = W[wcur ^ 1].p;
    w_blocked = rs;
    times_ht(X.p, Hp, true, done);                                         // :100 HH', :101 XH' (one launch)
    w_blocked = false;
    if (rs) scatter_w_numerator(o.update_H != 0, done);                    // sharded: numerator rows of this rank + summed HH'
    else allreduce_w_side(o.update_H != 0, done);
    spd_factor(gramH_p, (T)o.lambda_w, Uinv, "potrf_HHt", "trtri_HHt", done);       // :100 adddiag!, :102 potrf!
    // :102 potri! + copytri! + mul!, :103 projectnn!; sharded: rows of W are independent, this rank forms ITS Pc rows
    if (rs) {
        spd_solve_right(Uinv, invA, numW_p + row0, Wn + row0, Pc, true, done);
        stats_w_rows(Wn, Wo, done);
        gather_w_rows(Wn, true, done);
    } else {
        spd_solve_right(Uinv, invA, numW_p, Wn, P, true, done);
        stats_w(Wn, Wo, done);
    }
    wcur ^= 1;
}

}  // namespace nmfx
