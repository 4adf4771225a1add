// Development micro-benchmark for the Cholesky kernels (not part of the product build).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I nmf.jl_amd/csrc scripts/kbench/potrf_bench.hip -o /tmp/potrf_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#define NMFX_POTRF_TIMING 1
namespace nmfx { __device__ long long nmfx_potrf_dbg[2048]; }
#include "chol.hpp"
using namespace nmfx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename T> int run(int k) {
    const int K = (k + 127) / 128 * 128;
    std::vector<T> A((size_t)K * K, 0), G((size_t)k * k);
    srand(1);
    for (auto &g : G) g = (T)(rand() / (double)RAND_MAX);
    for (int i = 0; i < k; ++i) for (int j = 0; j < k; ++j) {
        double s = 0; for (int l = 0; l < k; ++l) s += (double)G[l + (size_t)i * k] * G[l + (size_t)j * k];
        A[i + (size_t)j * K] = (T)(s + (i == j ? 1.0 : 0.0));
    }
    T *dA, *dU, *dInv; Ctrl *ctrl;
    CK(hipMalloc(&dA, A.size() * sizeof(T))); CK(hipMalloc(&dU, A.size() * sizeof(T))); CK(hipMalloc(&dInv, A.size() * sizeof(T)));
    CK(hipMalloc(&ctrl, sizeof(Ctrl))); CK(hipMemset(ctrl, 0, sizeof(Ctrl)));
    const size_t kp0 = (size_t)(k + 31) / 32 * 32, kps = kp0 > 64 ? kp0 - 32 : 32;
    const size_t lds = (size_t)(32 * 32 + 32 * kps) * sizeof(T);
    const size_t lds_tri = (size_t)((k + 31) / 32 + 4) * 1024 * sizeof(T);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&trtri_offdiag_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tri));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&potrf_upper_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best_p = 1e9, best_t = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemcpy(dU, A.data(), A.size() * sizeof(T), hipMemcpyHostToDevice));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((potrf_upper_kernel<T>), dim3(1), dim3(1024), lds, 0, dU, (int64_t)K, k, ctrl, 3, (T *)nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best_p = std::min(best_p, ms);
        CK(hipMemset(dInv, 0, A.size() * sizeof(T)));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((trtri_diag_kernel<T>), dim3((k + 31) / 32), dim3(64), 0, 0, dU, dInv, (int64_t)K, k, (const int *)nullptr);
        hipLaunchKernelGGL((trtri_offdiag_kernel<T>), dim3((k + 31) / 32), dim3(256), lds_tri, 0, dU, dInv, (int64_t)K, k, (k + 31) / 32 + 1, (const int *)nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); best_t = std::min(best_t, ms);
    }
    std::vector<T> U(A.size()), Inv(A.size());
    CK(hipMemcpy(U.data(), dU, A.size() * sizeof(T), hipMemcpyDeviceToHost));
    CK(hipMemcpy(Inv.data(), dInv, A.size() * sizeof(T), hipMemcpyDeviceToHost));
    // check U'U = A and U*Uinv = I
    double e1m = 0, e2m = 0;
    for (int i = 0; i < k; ++i) for (int j = i; j < k; ++j) {
        double s = 0; for (int l = 0; l <= i; ++l) s += (double)U[l + (size_t)i * K] * U[l + (size_t)j * K];
        e1m = std::max(e1m, std::fabs(s - (double)A[i + (size_t)j * K]) / std::fabs((double)A[i + (size_t)j * K]));
        double t = 0; for (int l = i; l <= j; ++l) t += (double)U[i + (size_t)l * K] * Inv[l + (size_t)j * K];
        e2m = std::max(e2m, std::fabs(t - (i == j ? 1.0 : 0.0)));
    }
    Ctrl h; CK(hipMemcpy(&h, ctrl, sizeof h, hipMemcpyDeviceToHost));
    if (k == 256 && sizeof(T) == 4) {
        long long dbg[1024]; CK(hipMemcpyFromSymbol(dbg, HIP_SYMBOL(nmfx::nmfx_potrf_dbg), sizeof dbg));
        for (int b = 0; b < 8; ++b)
            printf("  step %d: stage+diag %lld  sync %lld  panel %lld  wb+trail %lld  sync %lld (cycles)\n", b, dbg[b*8+1]-dbg[b*8+0], dbg[b*8+2]-dbg[b*8+1],
                   dbg[b*8+3]-dbg[b*8+2], dbg[b*8+4]-dbg[b*8+3], dbg[b*8+5]-dbg[b*8+4]);
    }
    printf("k=%4d %s: potrf %.1f us  trtri %.1f us  |U'U-A|rel %.2e  |U*Uinv-I| %.2e  status %d\n", k, sizeof(T) == 4 ? "f32" : "f64",
           best_p * 1e3, best_t * 1e3, e1m, e2m, h.status);
    return 0;
}
// the register-resident kernel (potrf_reg_kernel): adddiag! and the diagonal blocks' inverses fused; checked as above
template <typename T, int NBLK> int run_reg(int k, double lambda) {
    const int K = NBLK * 32;
    std::vector<T> A((size_t)K * K, 0), G((size_t)k * k);
    srand(2);
    for (auto &g : G) g = (T)(rand() / (double)RAND_MAX);
    for (int i = 0; i < k; ++i) for (int j = 0; j < k; ++j) {
        double s = 0; for (int l = 0; l < k; ++l) s += (double)G[l + (size_t)i * k] * G[l + (size_t)j * k];
        A[i + (size_t)j * K] = (T)s;
    }
    T *dA, *dU, *dInv; Ctrl *ctrl;
    CK(hipMalloc(&dA, A.size() * sizeof(T))); CK(hipMalloc(&dU, A.size() * sizeof(T))); CK(hipMalloc(&dInv, A.size() * sizeof(T)));
    CK(hipMalloc(&ctrl, sizeof(Ctrl))); CK(hipMemset(ctrl, 0, sizeof(Ctrl)));
    const size_t lds = PotrfReg<T, NBLK>::lds_bytes();
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&potrf_reg_kernel<T, NBLK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best[2] = {1e9f, 1e9f};
    for (int with_inv = 0; with_inv < 2; ++with_inv)
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemcpy(dU, A.data(), A.size() * sizeof(T), hipMemcpyHostToDevice));
            CK(hipMemset(dInv, 0, A.size() * sizeof(T)));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((potrf_reg_kernel<T, NBLK>), dim3(1), dim3(512), lds, 0, dU, (int64_t)K, k, (T)lambda, with_inv ? dInv : (T *)nullptr, ctrl, 3);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best[with_inv] = std::min(best[with_inv], ms);
        }
    std::vector<T> U(A.size()), Inv(A.size());
    CK(hipMemcpy(U.data(), dU, A.size() * sizeof(T), hipMemcpyDeviceToHost));
    CK(hipMemcpy(Inv.data(), dInv, A.size() * sizeof(T), hipMemcpyDeviceToHost));
    double e1m = 0, e2m = 0, e3m = 0;
    for (int i = 0; i < k; ++i) for (int j = i; j < k; ++j) {
        double s = 0; for (int l = 0; l <= i; ++l) s += (double)U[l + (size_t)i * K] * U[l + (size_t)j * K];
        const double a = (double)A[i + (size_t)j * K] + (i == j ? lambda : 0.0);
        e1m = std::max(e1m, std::fabs(s - a) / std::fabs(a));
        if (i / 32 == j / 32) {   // diagonal blocks of the inverse: U_bb * Dinv_bb = I
            double t = 0; for (int l = i; l <= j; ++l) t += (double)U[i + (size_t)l * K] * Inv[l + (size_t)j * K];
            e2m = std::max(e2m, std::fabs(t - (i == j ? 1.0 : 0.0)));
        }
    }
    for (int i = 0; i < K; ++i) for (int j = 0; j < i; ++j)     // the strictly lower triangle stays what it was
        e3m = std::max(e3m, std::fabs((double)U[i + (size_t)j * K] - (double)A[i + (size_t)j * K]));
    if (k == 256 && sizeof(T) == 4 && lambda == 0.5) {   // phases of the last launch (with the inverses), wave by wave: cycles
        static long long dbg[2048]; CK(hipMemcpyFromSymbol(dbg, HIP_SYMBOL(nmfx::nmfx_potrf_dbg), sizeof dbg));
        for (int b = 0; b <= NBLK; ++b)
            for (int w = 0; w < 8; w += (b == 0 ? 1 : 7)) {
                const long long *d = dbg + 256 + (b * 8 + w) * 8;
                printf("  step %d wave %d: A %lld  wait %lld  B %lld  wait %lld  C %lld (copy %lld)\n", b, w, d[1] - d[0], d[2] - d[1], d[3] - d[2], d[4] - d[3], d[5] - d[4], w == 7 ? d[6] - d[4] : 0);
            }
    }
    Ctrl h; CK(hipMemcpy(&h, ctrl, sizeof h, hipMemcpyDeviceToHost));
    printf("k=%4d %s reg<%d>: potrf %.1f us  potrf + diagonal inverses %.1f us  |U'U-A|rel %.2e  |U_bb*Dinv_bb-I| %.2e  lower untouched %.1e  status %d  (lds %zu B)\n", k,
           sizeof(T) == 4 ? "f32" : "f64", NBLK, best[0] * 1e3, best[1] * 1e3, e1m, e2m, e3m, h.status, lds);
    return 0;
}
int main() {
    run<float>(70); run<float>(256); run<double>(256); run<float>(512); run<double>(512);
    run_reg<float, 8>(256, 0.5); run_reg<float, 8>(256, 0.0); run_reg<float, 8>(230, 0.5); run_reg<float, 6>(192, 0.25); run_reg<float, 4>(70, 0.5); run_reg<float, 2>(5, 1.0);
    run_reg<float, 4>(128, 0.5); run_reg<double, 4>(128, 0.5); run_reg<double, 4>(100, 0.0); run_reg<double, 2>(64, 0.5);
    return 0;
}
