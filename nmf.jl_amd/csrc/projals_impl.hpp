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
template <typename T> void Solver<T>::spd_factor(T *A, T lambda, T *Uinv, const char *tag_potrf, const char *tag_trtri, const int *done, T *Tm) {
    const size_t kk = (size_t)K * K;
    // potrf: 32 x 32 diagonal block + 32 x kp row panel in LDS (kp = k rounded up to 32)
    const size_t lds32 = potrf_lds_bytes();
    // trtri: up to `nfit` finished tiles of a block column + 4 partial tiles in LDS (96 KiB at most, so that the workgroup still fits
    // beside a GEMM block of the co-resident product); further tiles are read back from global memory (chol.hpp)
    const int nblk_t = (int)((k + 31) / 32);
    const int nfit = std::max(1, (int)((96 * 1024) / (1024 * sizeof(T))) - 4);
    const size_t lds_tri = (size_t)(std::min(nblk_t, nfit) + 4) * 1024 * sizeof(T);
    const bool gpanel = lds32 > 160 * 1024;   // the row panel no longer fits one workgroup's LDS: keep it in a global scratch buffer
    if (gpanel) potrf_panel.ensure(lds32 / sizeof(T));
    // register-resident factorisation (chol.hpp: potrf_reg_kernel): adddiag! and the diagonal blocks' inverses in the same launch
    // (Float64 keeps the LDS-panel kernel when the factorisation shares its CU with a block of the product -- potrf_nt == 512: its
    // register-resident form needs 256 registers per lane and would not be placed beside that block)
    const bool reg = potrf_reg_ok() && !(sizeof(T) == 8 && potrf_nt == 512);
    timed(tag_potrf, (double)k * k * k / 3.0, 0.0, [&] {
        if (reg) {
            // Uinv must be zero outside the blocks this launch and trtri write.  In stream order: zeroed here.  Under a product (potrf_nt
            // == 512): zeroed ONCE per solve by iterate() -- the 64-block memset launch waited 363 us on average for block slots on the
            // factorisation stream, in front of a potrf that takes 75-340 us -- and it stays zero: both writers only touch their own blocks
            if (potrf_nt != 512) HIP_TRY(hipMemsetAsync(Uinv, 0, kk * sizeof(T), stream));
            auto go = [&](auto nblk) {
                constexpr int NBLK = decltype(nblk)::value;
                const size_t lds = PotrfReg<T, NBLK>::lds_bytes();
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&potrf_reg_kernel<T, NBLK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL((potrf_reg_kernel<T, NBLK>), dim3(1), dim3(512), lds, stream, A, K, (int)k, lambda, Uinv, ctrl, (int)NMFX_ERR_NOT_POSDEF);
            };
            const int nblk = (int)(K / 32);
            if (nblk == 2) go(std::integral_constant<int, 2>{});
            else if (nblk == 4) go(std::integral_constant<int, 4>{});
            else if constexpr (sizeof(T) == 4) {
                if (nblk == 6) go(std::integral_constant<int, 6>{});
                else go(std::integral_constant<int, 8>{});
            }
            HIP_TRY(hipGetLastError());
            return;
        }
        if (lambda != (T)0)   // adddiag! skips lambda == 0 (src/utils.jl:18)
            hipLaunchKernelGGL(adddiag_kernel<T>, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, stream, A, K, (int)k, lambda, done);
        if (gpanel) {
            hipLaunchKernelGGL((potrf_upper_kernel<T, true>), dim3(1), dim3(potrf_nt), (size_t)(32 * 32 + 64) * sizeof(T), stream, A, K, (int)k, ctrl,
                               (int)NMFX_ERR_NOT_POSDEF, potrf_panel.p);
        } else {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&potrf_upper_kernel<T, false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds32));
            hipLaunchKernelGGL((potrf_upper_kernel<T, false>), dim3(1), dim3(potrf_nt), lds32, stream, A, K, (int)k, ctrl, (int)NMFX_ERR_NOT_POSDEF,
                               (T *)nullptr);
        }
        HIP_TRY(hipGetLastError());
    });
    timed(tag_trtri, (double)k * k * k / 3.0, 0.0, [&] {
        const unsigned nblk = (unsigned)((k + 31) / 32);
        if (!reg) {
            HIP_TRY(hipMemsetAsync(Uinv, 0, kk * sizeof(T), stream));
            hipLaunchKernelGGL((trtri_diag_kernel<T>), dim3(nblk), dim3(64), 0, stream, A, Uinv, K, (int)k, done);
        }
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&trtri_offdiag_kernel<T>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tri));
        if (Tm != nullptr && strip_ok() && defer_pack) {
            // (enqueue_projals launches the pack itself, on the main stream behind the product: see there)
        } else if (Tm != nullptr && strip_ok()) {   // ... packed block by block in the order potrs_strip_kernel's sweeps consume them
            // (sharing its CUs with a product the pack's 288 one-trip blocks trickle through the 8 half-empty CUs in 130-170 us, against 4 us
            // alone; 16 blocks that fit them at once were measured SLOWER, 390 us: co-resident, every instruction is ~10x slower, and then
            // each thread loops over 18 elements)
            const int64_t pack_blocks = std::min<int64_t>((strip_pack_elems((int)(K / 32)) + 255) / 256, 1024);
            hipLaunchKernelGGL((potrs_strip_pack_kernel<T>), dim3((unsigned)pack_blocks), dim3(256), 0, stream, A, Uinv, Tm, K, (int)k, (int)(K / 32), done);
        }
        else if (Tm != nullptr)   // left solve by substitution (potrs!): only the diagonal blocks' inverses are needed, packed with U and U' (chol.hpp)
            hipLaunchKernelGGL((potrs_prep_kernel<T>), dim3((unsigned)std::min<int64_t>((K * K + 255) / 256, 1024)), dim3(256), 0, stream, A, Uinv, Tm, K, (int)k, (int)K, done);
        else
            hipLaunchKernelGGL((trtri_offdiag_kernel<T>), dim3(nblk), dim3(256), lds_tri, stream, A, Uinv, K, (int)k, nfit, done);
        HIP_TRY(hipGetLastError());
    });
}

// potrs! (src/utils.jl:69) after spd_factor(..., Tm): out = inv(A) B by the two blocked triangular substitutions of potrs_panel_kernel;
// clamp = projectnn!; old != nullptr: stop_condition's sums against `old` into stat_part (finalised by stats_h_finalize(N / NB))
template <typename T> int Solver<T>::spd_solve_left_potrs(const T *Tm, const T *B, T *out, bool clamp, const T *old, const int *done) {
    if (strip_ok()) {
        const unsigned grid = (unsigned)(N / STRIP_COLS);
        double *part = old ? stat_part.p : (double *)nullptr;
        timed("potrs_clampH", 2.0 * (double)K * K * N, 3.0 * K * N * sizeof(T), [&] {
            auto go = [&](auto nblk) {
                hipLaunchKernelGGL((potrs_strip_kernel<T, decltype(nblk)::value>), dim3(grid), dim3(64 * STRIP_WAVES), 0, stream, Tm, B, 1, (int64_t)0, K, out, old, clamp ? 1 : 0, part,
                                   (int)K, done);
            };
            const int nblk = (int)(K / 32);
            if (nblk == 2) go(std::integral_constant<int, 2>{});
            else if (nblk == 4) go(std::integral_constant<int, 4>{});
            else if (nblk == 6) go(std::integral_constant<int, 6>{});
            else if (nblk == 8) go(std::integral_constant<int, 8>{});
            else if constexpr (sizeof(T) == 4) {
                if (nblk == 10) go(std::integral_constant<int, 10>{});
                else if (nblk == 12) go(std::integral_constant<int, 12>{});
                else if (nblk == 14) go(std::integral_constant<int, 14>{});
                else go(std::integral_constant<int, 16>{});
            }
            HIP_TRY(hipGetLastError());
        });
        return (int)grid;
    }
    constexpr int NB = POTRS_NB;
    const size_t s_bytes = (size_t)K * (NB + 1) * sizeof(T), tp_bytes = (size_t)32 * K * sizeof(T);
    const bool dbuf = s_bytes + 2 * tp_bytes <= (size_t)160 * 1024;
    const size_t lds = s_bytes + (dbuf ? 2 : 1) * tp_bytes;
    timed("potrs_clampH", 2.0 * (double)K * K * N, 3.0 * K * N * sizeof(T), [&] {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&potrs_panel_kernel<T, NB, POTRS_NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((potrs_panel_kernel<T, NB, POTRS_NT>), dim3((unsigned)(N / NB)), dim3(POTRS_NT), lds, stream, Tm, K, B, 1, (int64_t)0, K, out, old, (int)K, clamp ? 1 : 0,
                           dbuf ? 1 : 0, old ? stat_part.p : (double *)nullptr, (int)K, done);
        HIP_TRY(hipGetLastError());
    });
    return (int)(N / NB);
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
    work[2].ensure(potrs_pack_elems());
    Ctrl init;
    std::memset(&init, 0, sizeof init);
    HIP_TRY(hipMemcpyAsync(ctrl, &init, sizeof init, hipMemcpyHostToDevice, stream));
    // the SPD matrix goes where projals keeps its Gram (gramW_p for the left solve, gramH_p for the right one), the other
    // operand where the numerator lives; everything zero-padded
    T *G = right ? gramH_p : gramW_p;
    HIP_TRY(hipMemsetAsync(G, 0, kk * sizeof(T), stream));
    HIP_TRY(hipMemcpy2DAsync(G, K * sizeof(T), right ? B_host : A_host, k * sizeof(T), k * sizeof(T), k, hipMemcpyHostToDevice, stream));
    const bool subst = !right && potrs_route_ok();
    spd_factor(G, (T)lambda, work[1].p, "potrf", "trtri", nullptr, subst ? work[2].p : (T *)nullptr);
    have_F = false;   // the factor buffers are scratch for this call
    if (!right) {     // x <- inv(A) x, x is k x n
        HIP_TRY(hipMemsetAsync(numH_p, 0, (size_t)K * N * sizeof(T), stream));
        HIP_TRY(hipMemcpy2DAsync(numH_p, K * sizeof(T), B_host, k * sizeof(T), k * sizeof(T), n, hipMemcpyHostToDevice, stream));
        if (subst) spd_solve_left_potrs(work[2].p, numH_p, H[0].p, clamp, nullptr, nullptr);
        else spd_solve_left(work[1].p, numH_p, work[0].p, H[0].p, clamp, nullptr);
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
    const int *done = done_flag();
    const size_t kk = (size_t)K * K;
    work[0].ensure((size_t)K * N);   // Y = Uinv' * W'X
    work[1].ensure(kk);              // Uinv
    work[2].ensure(potrs_pack_elems());   // inv(HH' + lw I); on the H side the packed factor of the substitution route
    T *Y = work[0].p, *Uinv = work[1].p, *invA = work[2].p;
    const bool rs = row_sharded();
    // The factorisations run UNDER the big products.  Each is one workgroup (potrf) and a few small ones (trtri, potri) whose
    // 32-step dependency chains take 265 us per side at k = 256 -- 19 % of an iteration when they sit between the products.
    // A big product is ONE wave of 2 blocks per CU that stay resident for the whole launch and own every register of their
    // SIMDs, so a second stream never gets a workgroup placed beside them (measured in round 2: 2.76 ms either way; CU masks do
    // not help either: the hardware deals workgroups to shader engines round-robin whatever their CU count, so a mask that
    // takes one CU from one engine slows the whole product by 1/8 -- scripts/kbench/cu_mask_probe.hip).  What works: the Gram
    // by its own small launch first, then the product launched `chol_slots` blocks SHORT (plan_short_grid: the missing items
    // ride as tail pieces, +1.6 % work per block) -- that leaves chol_slots half-empty CUs -- and the factorisation on a
    // high-priority side stream with a 512-thread Cholesky workgroup (2 waves per SIMD x 128 registers = the free half of a
    // CU).  Sharing its CU with a GEMM block the chain runs 4x slower (potrf 650 us, trtri 400 us), which still fits under the
    // 1.02 ms product: 2.75 -> 2.36 ms per iteration at 16384 x 16384, k = 256.  Replicated-W multi-GPU mode keeps the serial
    // order (its Gram travels inside the one packed all-reduce that follows the product).
    // (round 6: only under a product long enough to cover most of the chain.  Sharing its CU the register-resident potrf takes
    // 0.53-0.62 ms at k = 256 -- seven times its stand-alone 77 us: it is bound by instruction issue, which the product's waves
    // contend for -- so under a short product the chain is the iteration (8192 x 4096: 0.58 ms under the 115 us products, 0.54 in
    // stream order); from ~200 us on hiding wins: solver.hpp, chol_under_min_us)
    // (end of round 6 -- what the two paragraphs above describe is NOT co-residency.  A workgroup of another kernel executes nothing on a
    // CU that holds a block of back-to-back MFMAs until that block has left, whatever its priority (scripts/kbench/coresident_probe.hip;
    // cycle stamps inside potrf_reg_kernel: 67 us from its first to its last instruction in every form, the wait is in front of the
    // first).  The short grid works because the LONE blocks of its chol_slots half-empty CUs run their half-length items with the matrix
    // pipes to themselves and are gone at HALF TIME: from then on the chain has 8 empty CUs, one per XCD, and runs at stand-alone speed --
    // potrf ends at half time + 66 us.  Hence the crossover below: a product whose half time is shorter than the chain cannot hide it.
    // DESIGN.md section 3.2, item (6).)
    const double prod_us = 2.0 * (double)P * (double)N * (double)K / (sizeof(T) == 4 ? 150e6 : 70e6);
    const bool under = chol_slots > 0 && !use_bf16x3() && K % 128 == 0 && (!sharded() || rs) && prod_us >= chol_under_min_us;
    if (under) ensure_fstream();
    // H solve by triangular substitution (potrs!, the reference's route) instead of Uinv (Uinv' B): opt-in for the ITERATION
    // (NMFX_POTRS=1) -- a panel's two sweeps are a chain of 2 K / 32 dependent block steps that one workgroup per CU cannot overlap
    // with anything: 79 us at 16384 columns, k = 256, Float32 against 55 us for the two k x k x n products (scripts/kbench/potrs_bench.hip;
    // Float64, k = 128: 26 against 28 us).  nmfx_pdsolve, the exported pdsolve!, always takes the substitution route.
    // nmfx_opts.h_solve picks the route of the H solve (NMFX_POTRS=1 in the environment still forces substitution: A/B runs)
    // (round 5: with the strip kernel the substitution is the faster route AND the reference's, so AUTO takes it wherever that kernel exists)
    // (an EXPLICIT request for the substitution route that cannot be honoured -- k beyond the strip kernel and a panel that does not
    // fit the LDS -- is an error, not a silent change of arithmetic: the caller asked for the reference's potrs! and must be able to
    // tell which route ran)
    if (o.h_solve == NMFX_HSOLVE_POTRS && !potrs_route_ok())
        throw StatusError{NMFX_ERR_UNSUPPORTED, "h_solve = NMFX_HSOLVE_POTRS: the substitution route does not exist for this k (use NMFX_HSOLVE_AUTO or NMFX_HSOLVE_PRODUCT)"};
    const bool subst = (o.h_solve == NMFX_HSOLVE_POTRS || (o.h_solve == NMFX_HSOLVE_AUTO && (potrs_iter || strip_ok()))) && potrs_route_ok();
    bool potri_on_main = false;
    auto factor_under = [&](T *G, T lambda, const char *t1, const char *t2, bool with_potri) {
        HIP_TRY(hipEventRecord(ev_fork, stream));
        HIP_TRY(hipStreamWaitEvent(fstream, ev_fork, 0));
        {
            // every launch helper targets `stream`: point it at the side stream for the duration (restored on any exit path)
            struct Swap {
                Solver<T> &s;
                explicit Swap(Solver<T> &s_) : s(s_) { std::swap(s.stream, s.fstream); s.potrf_nt = 512; }
                ~Swap() { std::swap(s.stream, s.fstream); s.potrf_nt = 1024; }
            } on_side(*this);
            spd_factor(G, lambda, Uinv, t1, t2, done, (!with_potri && subst) ? invA : (T *)nullptr);
            if (with_potri && !potri_on_main) {   // potri! + copytri! (src/utils.jl:79-80) belong to the factorisation, not to the product
                EpiStore<T> e1{invA, K, 0, nullptr};
                gemm<KSTRIDED, KSTRIDED>("gemm_potri", Uinv, K, K, Uinv, K, K, K, 1, true, e1, done, 2.0 * K * K * sizeof(T));
            }
        }
        HIP_TRY(hipEventRecord(ev_join, fstream));
    };
    if (o.update_H) {
        const T *Wp = W[wcur].p;
        const T *Ho = H[hcur].p;
        T *Hn = H[hcur ^ 1].p;
        if (under) {
            // :92 W'W (W is replicated: no exchange) -- or, from the second iteration of the fused row-sharded step on, already there:
            // the sum over the ranks of W_g'W_g of their own new rows, which travelled with the all-gather of W
            if (!gramw_sharded_valid) gram_w_only(Wp, done);
            // (round 6: with the register-resident potrf the diagonal blocks' inverses come out of the factorisation launch, and the pack
            // of the factor for the strip kernel -- 4 us alone, 130-370 us when its 288 blocks have to find slots under the product --
            // runs on the main stream behind the product instead: the chain under W'X is the potrf alone, ~530 of the product's 930 us)
            defer_pack = subst && strip_ok() && potrf_reg_ok();
            try { factor_under(gramW_p, (T)o.lambda_h, "potrf_WtW", "trtri_WtW", false); } catch (...) { defer_pack = false; throw; }   // :92 adddiag!, :94 potrf!
            const bool packed_later = defer_pack;
            defer_pack = false;
            short_grid = true;
            try { wt_times(Wp, X.p, false, done); } catch (...) { short_grid = false; throw; }   // :93 H <- W'X
            short_grid = false;
            HIP_TRY(hipStreamWaitEvent(stream, ev_join, 0));
            if (packed_later)
                timed("pack_WtW", 0.0, 2.0 * (double)strip_pack_elems((int)(K / 32)) * sizeof(T), [&] {
                    hipLaunchKernelGGL((potrs_strip_pack_kernel<T>), dim3((unsigned)std::min<int64_t>((strip_pack_elems((int)(K / 32)) + 255) / 256, 1024)), dim3(256), 0, stream,
                                       gramW_p, Uinv, invA, K, (int)k, (int)(K / 32), done);
                    HIP_TRY(hipGetLastError());
                });
        } else {
            wt_times(Wp, X.p, true, done);                                     // :92 W'W, :93 H <- W'X (one launch)
            spd_factor(gramW_p, (T)o.lambda_h, Uinv, "potrf_WtW", "trtri_WtW", done, subst ? invA : (T *)nullptr);
        }
        if (subst) {   // :94 potrs! (two blocked triangular substitutions), :95 projectnn!, stop_condition's sums over H -- one launch
            const int chunks = spd_solve_left_potrs(invA, numH_p, Hn, true, Ho, done);
            h_stat_chunks = chunks;
            if (!stats_fuse_ok(o)) stats_h_finalize(chunks, done);   // (one GPU, nothing tracked: finalised by the launch behind the W update, as MultUpdate-MSE's)
        } else {   // :94 potrs! as Uinv * (Uinv' * B), :95 projectnn! and stop_condition's sums over H in the second product's epilogue
            EpiStore<T> e1{Y, K, 0, nullptr};
            gemm<KCONTIG, KCONTIG>("gemm_UinvtB", numH_p, K, N, Uinv, K, K, K, 1, true, e1, done, 2.0 * K * N * sizeof(T));
            EpiClampStats<T> e2{Ho, Hn, K, stat_part.p, (int)K};
            gemm<KCONTIG, KSTRIDED>("gemm_UinvY_clampH", Y, K, N, Uinv, K, K, K, 1, true, e2, done, 3.0 * K * N * sizeof(T));
            h_stat_chunks = last_tiles_r;
            if (!stats_fuse_ok(o)) stats_h_finalize(last_tiles_r, done);
        }
        hcur ^= 1;
    }
    const T *Hp = H[hcur].p;
    const T *Wo = W[wcur].p;
    T *Wn = W[wcur ^ 1].p;
    if (under) {
        // (H' for the transposed-image product first: in front of the fork, so that its 1024-block pass does not run beside the potrf's start)
        const T *HtP = xht_images ? ht_for(Hp, done, /*sharded_too=*/true) : nullptr;   // (row-sharded step too: X' and H' of the rank's own columns)
        gram_h_only(Hp, done);                                                 // :100 HH' of this rank's columns ...
        if (rs) timed("all_reduce_HHt", 0.0, (double)kk * sizeof(T), [&] { comm->all_reduce(gramH_p, kk, CT, false, stream); });   // ... summed
        // (with XH' on the transposed images potri!'s product runs on the main stream behind XH': 13 us there, 156-252 us beside the product)
        potri_on_main = HtP != nullptr;
        factor_under(gramH_p, (T)o.lambda_w, "potrf_HHt", "trtri_HHt", true);  // :100 adddiag!, :102 potrf!, potri!, copytri!
        w_blocked = rs;
        // (round 6, one GPU: XH' on the transposed images -- the contraction-contiguous kernel with its k-loop unrolled, as W'X runs:
        // 986 -> 931 us.  Round 5 measured the iteration SLOWER with it, and so did three attempts of this round -- until H's transposition
        // pass moved in FRONT of the fork (above): issued behind it, its 1024 blocks ran beside the start of the potrf on the other stream
        // and the chain lost more than the product gained.  With potri's product behind XH' on the main stream (13 us there, 156-252 beside
        // the product) the chain under XH' is potrf + trtri, ~630 of the product's 931 us: 2.10-2.11 -> 2.06-2.07 ms per iteration.)
        short_grid = true;
        try { times_ht(X.p, Hp, false, done, false, HtP); } catch (...) { short_grid = false; w_blocked = false; throw; }   // :101 XH'
        short_grid = false;
        w_blocked = false;
        if (rs && rs_fused_enabled) {
            // The row-sharded W side with the launches around its two collectives fused, as MultUpdate-MSE has it (solver_impl.hpp):
            // the product reads the rank's rows of the numerator STRAIGHT from the reduce-scatter's output and writes the new rows
            // STRAIGHT into the rank's chunk of the all-gather buffer (no piece_to_rows / rows_to_piece / gathered_to_full launches),
            // W'W for the next H solve comes from the rank's own new rows (2 Pc k^2 instead of 2 p k^2 on every rank) and travels with
            // the in-place all-gather, and one launch unpacks W and takes stop_condition's column sums over all rows.
            timed("comm_reduce_scatter_numW", 0.0, (double)(P * K) * sizeof(T), [&] {
                comm->group_start();
                comm->reduce_scatter(numW_p, rs_out.p, (size_t)Pc * K, CT, stream);
                if (o.update_H) comm->all_reduce(hstat.p, (size_t)2 * K, CT_F64, false, stream);
                comm->group_end();
            });
            HIP_TRY(hipStreamWaitEvent(stream, ev_join, 0));
            if (potri_on_main) {             // (potri! behind the product instead of beside it: see factor_under)
                EpiStore<T> e1{invA, K, 0, nullptr};
                gemm<KSTRIDED, KSTRIDED>("gemm_potri", Uinv, K, K, Uinv, K, K, K, 1, true, e1, done, 2.0 * K * K * sizeof(T));
            }
            const size_t chunk = (size_t)Pc * K * sizeof(T);
            T *mine = reinterpret_cast<T *>(ag_recv.p + (size_t)rank * chunk);
            EpiClampStore<T> e2{mine, Pc};                                     // :102 mul!, :103 projectnn!
            gemm<KSTRIDED, KSTRIDED>("gemm_XHtInv_clampW", invA, K, K, rs_out.p, Pc, Pc, K, 1, false, e2, done, 2.0 * Pc * K * sizeof(T));
            if (o.update_H) {
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
                if (o.update_H) comm->all_reduce(gramW_p, kk, CT, false, stream);
                comm->group_end();
            });
            gramw_sharded_valid = o.update_H != 0;
            const bool fuse_check = o.track_objective == 0;
            const int cpp = (int)std::max<int64_t>(1, std::min<int64_t>(64 / nranks, Pc / 1024));
            timed("gather_W_stats", 0.0, 3.0 * P * K * sizeof(T), [&] {
                hipLaunchKernelGGL(gather_stats_kernel<T>, dim3((unsigned)(nranks * cpp), (unsigned)K), dim3(256), 0, stream, Wn, Wo, ag_recv.p, chunk, P, Pc, cpp, (int)K,
                                   stat_part.p, done);
                hipLaunchKernelGGL(stats_check_kernel<T>, dim3(1), dim3(256), 0, stream, stat_part.p, nranks * cpp, (int)K, wstat.p, ctrl,
                                   o.update_H ? hstat.p : (const double *)nullptr, (int)k, (T)o.tol, t, fuse_check ? 1 : 0, done);
                HIP_TRY(hipGetLastError());
            });
            check_fused = fuse_check;
            wcur ^= 1;
            return;
        }
        if (rs) scatter_w_numerator(o.update_H != 0, done, /*with_tail=*/false);
        HIP_TRY(hipStreamWaitEvent(stream, ev_join, 0));
        if (potri_on_main) {             // (see above: potri! behind the product instead of beside it)
            EpiStore<T> e1{invA, K, 0, nullptr};
            gemm<KSTRIDED, KSTRIDED>("gemm_potri", Uinv, K, K, Uinv, K, K, K, 1, true, e1, done, 2.0 * K * K * sizeof(T));
        }
        const int64_t r0 = rs ? row0 : 0, rows = rs ? Pc : P;
        EpiClampStore<T> e2{Wn + r0, P};                                       // :102 mul!, :103 projectnn!
        gemm<KSTRIDED, KSTRIDED>("gemm_XHtInv_clampW", invA, K, K, numW_p + r0, P, rows, K, 1, false, e2, done, 2.0 * rows * K * sizeof(T));
    } else {
        w_blocked = rs;
        times_ht(X.p, Hp, true, done);                                         // :100 HH', :101 XH' (one launch)
        w_blocked = false;
        if (rs) scatter_w_numerator(o.update_H != 0, done);                    // sharded: numerator rows of this rank + summed HH'
        else allreduce_w_side(o.update_H != 0, done);
        spd_factor(gramH_p, (T)o.lambda_w, Uinv, "potrf_HHt", "trtri_HHt", done);   // :100 adddiag!, :102 potrf!
        // :102 potri! + copytri! + mul!, :103 projectnn!; sharded: rows of W are independent, this rank forms ITS Pc rows
        spd_solve_right(Uinv, invA, numW_p + (rs ? row0 : 0), Wn + (rs ? row0 : 0), rs ? Pc : P, true, done);
    }
    if (rs) {
        stats_w_rows(Wn, Wo, done);
        gather_w_rows(Wn, true, done);
    } else if (stats_fuse_ok(o)) {
        stats_w_check_fused(Wn, Wo, o, t, done);
    } else {
        stats_w(Wn, Wo, done);
    }
    wcur ^= 1;
}

}  // namespace nmfx
