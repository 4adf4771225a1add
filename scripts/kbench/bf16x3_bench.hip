// Development probe: fp32 GEMM emulated with three bf16 MFMA products (a = hi + lo split; hi*hi + hi*lo + lo*hi), the
// "mixed-precision option" of SURVEY.md section 8f rank 4.  KCONTIG x KCONTIG, 128 x 128 tiles, split-K, plain store.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/kbench/bf16x3_bench.hip -o scripts/kbench/bf16x3_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BR = 128, BC = 128, BK = 32, NT = 256;
constexpr int PLANE = BR * BK * 2;          // bytes of one bf16 plane of one operand tile (8 KiB)
constexpr int OPER = 2 * PLANE;             // hi + lo
constexpr int STAGE = 2 * OPER;             // A + B  (32 KiB)

// LDS image of one bf16 plane: [chunk (8 k = 16 B)][row][16 B]; the row slot is permuted so that the fragment reads
// (32 consecutive rows, one chunk), the KCONTIG writes (one row, 4 chunks x 2 halves per 8 lanes) and the KSTRIDED
// micro-tile writes (rows 4 apart) are all (nearly) conflict-free:  slot = row ^ ((row >> 4) & 3) ^ (chunk << 2)
__device__ __forceinline__ int lds_off(int row, int chunk) { return chunk * (BR * 16) + ((row ^ ((row >> 4) & 3) ^ (chunk << 2)) << 4); }

// split 4 consecutive-k fp32 values into bf16 hi / lo quads
__device__ __forceinline__ void split4(const f32x4 v, bf16x4 &hi, bf16x4 &lo) {
    hi = __builtin_convertvector(v, bf16x4);
    const f32x4 hf = __builtin_convertvector(hi, f32x4);
    lo = __builtin_convertvector(v - hf, bf16x4);
}

__device__ __forceinline__ void load_tile(f32x4 (&r)[4], const float *base, int64_t ld, int64_t row0, int64_t k0, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = tid + NT * i;
        const int row = s >> 3, cpos = s & 7;          // 8 quads (32 k) per row
        r[i] = *reinterpret_cast<const f32x4 *>(base + (row0 + row) * ld + k0 + cpos * 4);
    }
}
__device__ __forceinline__ void store_tile(const f32x4 (&r)[4], char *oper, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = tid + NT * i;
        const int row = s >> 3, cpos = s & 7;
        bf16x4 hi, lo;
        split4(r[i], hi, lo);
        const int off = lds_off(row, cpos >> 1) + ((cpos & 1) << 3);
        *reinterpret_cast<bf16x4 *>(oper + off) = hi;
        *reinterpret_cast<bf16x4 *>(oper + PLANE + off) = lo;
    }
}
// KSTRIDED operand (element (row, k) at base[k*ld + row]): a thread owns one 4 x 4 micro-tile (4 consecutive rows x 4
// consecutive k): four 16-byte loads, transposed in registers at store time
__device__ __forceinline__ void load_tile_ks(f32x4 (&r)[4], const float *base, int64_t ld, int64_t row0, int64_t k0, int tid) {
    const int kq = tid >> 5, r4 = tid & 31;
#pragma unroll
    for (int ek = 0; ek < 4; ++ek) r[ek] = *reinterpret_cast<const f32x4 *>(base + (k0 + kq * 4 + ek) * ld + row0 + r4 * 4);
}
__device__ __forceinline__ void store_tile_ks(const f32x4 (&r)[4], char *oper, int tid) {
    const int kq = tid >> 5, r4 = tid & 31;
#pragma unroll
    for (int er = 0; er < 4; ++er) {
        const f32x4 v = {r[0][er], r[1][er], r[2][er], r[3][er]};
        bf16x4 hi, lo;
        split4(v, hi, lo);
        const int off = lds_off(r4 * 4 + er, kq >> 1) + ((kq & 1) << 3);
        *reinterpret_cast<bf16x4 *>(oper + off) = hi;
        *reinterpret_cast<bf16x4 *>(oper + PLANE + off) = lo;
    }
}
__device__ __forceinline__ bf16x8 frag(const char *plane, int rt, int ks, int lane) {
    const int r = rt + (lane & 31), c = 2 * ks + (lane >> 5);
    return *reinterpret_cast<const bf16x8 *>(plane + lds_off(r, c));
}

// D(r, c) = sum_k A(r, k) B(c, k);  slab s at D + s*R*C, element (r, c) at c + r*ldd
template <int KS>
__global__ __launch_bounds__(NT) void gemm_bf16x3(const float *A, const float *B, float *D, int64_t lda, int64_t ldb, int64_t ldd,
                                                  int tiles_r, int tiles_c, int tiles, int kchunk, int64_t slab_stride, int c_fastest) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int split = bid / tiles, trem = bid % tiles;
    const int tc = c_fastest ? trem % tiles_c : trem / tiles_r, tr = c_fastest ? trem / tiles_c : trem % tiles_r;
    const int64_t r0 = (int64_t)tr * BR, c0 = (int64_t)tc * BC, kbeg = (int64_t)split * kchunk;
    const int nk = kchunk / BK;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    f32x4 ra[4], rb[4];
    auto ld = [&](f32x4 (&r)[4], const float *base, int64_t l, int64_t row0, int64_t k0) {
        if constexpr (KS) load_tile_ks(r, base, l, row0, k0, tid); else load_tile(r, base, l, row0, k0, tid);
    };
    auto st = [&](const f32x4 (&r)[4], char *oper) {
        if constexpr (KS) store_tile_ks(r, oper, tid); else store_tile(r, oper, tid);
    };
    ld(ra, A, lda, r0, kbeg);
    ld(rb, B, ldb, c0, kbeg);
    st(ra, smem);
    st(rb, smem + OPER);
    if (nk > 1) { ld(ra, A, lda, r0, kbeg + BK); ld(rb, B, ldb, c0, kbeg + BK); }
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        const char *a_s = smem + (t & 1) * STAGE, *b_s = a_s + OPER;
        char *a_n = smem + ((t & 1) ^ 1) * STAGE, *b_n = a_n + OPER;
        if (t + 1 < nk) { st(ra, a_n); st(rb, b_n); }
        if (t + 2 < nk) {
            ld(ra, A, lda, r0, kbeg + (int64_t)(t + 2) * BK);
            ld(rb, B, ldb, c0, kbeg + (int64_t)(t + 2) * BK);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { ah[i] = frag(a_s, wr * 64 + i * 32, ks, lane); al[i] = frag(a_s + PLANE, wr * 64 + i * 32, ks, lane); }
#pragma unroll
            for (int j = 0; j < 2; ++j) { bh[j] = frag(b_s, wc * 64 + j * 32, ks, lane); bl[j] = frag(b_s + PLANE, wc * 64 + j * 32, ks, lane); }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    float *dst = D + (int64_t)split * slab_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int64_t r = r0 + wr * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                const int64_t c = c0 + wc * 64 + j * 32 + (lane & 31);
                dst[c + r * ldd] = acc[i][j][reg];
            }
}

template <int KS> void run(const char *name, int64_t R, int64_t C, int64_t Kd, int splits, int c_fastest, int reps) {
    float *A, *B, *D;
    CK(hipMalloc(&A, R * Kd * 4)); CK(hipMalloc(&B, C * Kd * 4)); CK(hipMalloc(&D, R * C * splits * 4));
    std::vector<float> hA((size_t)R * Kd), hB((size_t)C * Kd);
    const bool a_big = R >= C;
    for (auto &v : hA) { const double u = rand() / (double)RAND_MAX; v = (float)(a_big ? 64.0 * u : u / 16384.0); }
    for (auto &v : hB) { const double u = rand() / (double)RAND_MAX; v = (float)(a_big ? u / 16384.0 : 64.0 * u); }
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    const int64_t lda = KS ? R : Kd, ldb = KS ? C : Kd;
    const int tiles_r = (int)(R / BR), tiles_c = (int)(C / BC), tiles = tiles_r * tiles_c;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(gemm_bf16x3<KS>, dim3(tiles * splits), dim3(NT), 0, 0, A, B, D, lda, ldb, C, tiles_r, tiles_c, tiles,
                           (int)(Kd / splits), R * C, c_fastest);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (i > 1) best = std::min(best, ms);
    }
    CK(hipGetLastError());
    std::vector<float> hD((size_t)R * C * splits);
    CK(hipMemcpy(hD.data(), D, hD.size() * 4, hipMemcpyDeviceToHost));
    double maxrel = 0, maxrel32 = 0;
    for (int s = 0; s < 256; ++s) {
        const int64_t r = rand() % R, c = rand() % C;
        double ref = 0; float f32 = 0.f;
        for (int64_t k = 0; k < Kd; ++k) {
            const float a = KS ? hA[k * lda + r] : hA[r * lda + k], b = KS ? hB[k * ldb + c] : hB[c * ldb + k];
            ref += (double)a * (double)b; f32 += a * b;
        }
        double got = 0;
        for (int sp = 0; sp < splits; ++sp) got += hD[(size_t)sp * R * C + c + r * C];
        maxrel = std::max(maxrel, std::fabs(got - ref) / std::fabs(ref));
        maxrel32 = std::max(maxrel32, std::fabs((double)f32 - ref) / std::fabs(ref));
    }
    printf("%-22s R=%lld C=%lld K=%lld splits=%d: %.1f us  %.1f TF/s fp32-equivalent; max rel err vs fp64 %.2e (sequential fp32 loop: %.2e)\n",
           name, (long long)R, (long long)C, (long long)Kd, splits, best * 1e3, 2.0 * R * C * Kd / (best * 1e-3) / 1e12, maxrel, maxrel32);
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(D));
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    run<0>("TN (WtX) bf16x3", 16384, 256, 16384, 2, 1, reps);
    run<1>("NT (XHt) bf16x3", 256, 16384, 16384, 2, 0, reps);
    return 0;
}
