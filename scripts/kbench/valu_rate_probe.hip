// valu_rate_probe.hip -- issue rate of the VALU instructions on the greedy step's chain (gfx950), relative to v_fma_f32:
// 8 waves per SIMD, each running ITER x 8 independent copies of one instruction.  Prints ns per wave-instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 valu_rate_probe.hip -o valu_rate_probe && ./valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ITER = 4096;
#define BODY8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int OP> __global__ __launch_bounds__(256) void k(float *out, float seed) {
    float a[8]; double d[8]; float b[16];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x + i; d[i] = (double)a[i] * 1.0000001; }
    for (int i = 0; i < 16; ++i) b[i] = seed * i + 1.0f;
    const double c = 1.0000000001; const float cf = 1.0000001f;
    for (int it = 0; it < ITER; ++it) {
        if constexpr (OP == 0) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(cf));
            BODY8(S)
#undef S
        } else if constexpr (OP == 1) {
#define S(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
            BODY8(S)
#undef S
        } else if constexpr (OP == 2) {
#define S(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
            BODY8(S)
#undef S
        } else if constexpr (OP == 3) {
#define S(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c));
            BODY8(S)
#undef S
        } else if constexpr (OP == 4) {
            typedef float f2 __attribute__((ext_vector_type(2)));
#define S(i) { f2 &p = *reinterpret_cast<f2 *>(&b[2 * i]); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(p)); }
            BODY8(S)
#undef S
        } else if constexpr (OP == 5) {
#define S(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(cf) : );
            BODY8(S)
#undef S
        } else if constexpr (OP == 6) {
#define S(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            BODY8(S)
#undef S
        } else if constexpr (OP == 7) {
#define S(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(c));
            BODY8(S)
#undef S
        } else if constexpr (OP == 8) {   // dependent chain of f32 fma (latency)
            asm volatile("v_fma_f32 %0, %0, %1, %0\n\tv_fma_f32 %0, %0, %1, %0\n\tv_fma_f32 %0, %0, %1, %0\n\tv_fma_f32 %0, %0, %1, %0\n\t"
                         "v_fma_f32 %0, %0, %1, %0\n\tv_fma_f32 %0, %0, %1, %0\n\tv_fma_f32 %0, %0, %1, %0\n\tv_fma_f32 %0, %0, %1, %0" : "+v"(a[0]) : "v"(cf));
        } else if constexpr (OP == 9) {   // dependent chain cvt -> mul64 -> cvt (the greedy_div chain), 2 rounds + 2 filler = 8 instr
            asm volatile("v_cvt_f64_f32 %1, %0\n\tv_mul_f64 %1, %1, %2\n\tv_cvt_f32_f64 %0, %1\n\tv_cvt_f64_f32 %1, %0\n\tv_mul_f64 %1, %1, %2\n\tv_cvt_f32_f64 %0, %1\n\t"
                         "v_cvt_f64_f32 %1, %0\n\tv_mul_f64 %1, %1, %2" : "+v"(a[0]), "+v"(d[0]) : "v"(c));

        } else if constexpr (OP == 10) {
#define S(i) asm volatile("v_cmp_gt_f32 vcc, %1, %0\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(cf) : "vcc");
            BODY8(S)
#undef S
        } else if constexpr (OP == 11) {
#define S(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(cf) : );
            BODY8(S)
#undef S
        } else if constexpr (OP == 12) {
#define S(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
            BODY8(S)
#undef S
        } else if constexpr (OP == 13) {
#define S(i) asm volatile("v_cmp_gt_f32 vcc, %1, %0" : : "v"(a[i]), "v"(cf) : "vcc");
            BODY8(S)
#undef S
        } else if constexpr (OP == 14) {
#define S(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(cf));
            BODY8(S)
#undef S
        } else if constexpr (OP == 15) {
#define S(i) asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[0]));
            BODY8(S)
#undef S
        } else if constexpr (OP == 16) {
#define S(i) { int s_; asm volatile("v_readlane_b32 %0, %1, 63" : "=s"(s_) : "v"(a[i])); }
            BODY8(S)
#undef S
        } else if constexpr (OP == 17) {
            typedef float f2 __attribute__((ext_vector_type(2)));
#define S(i) { f2 &p = *reinterpret_cast<f2 *>(&b[2 * i]); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(p)); }
            BODY8(S)
#undef S
        } else if constexpr (OP == 18) {
#define S(i) { unsigned long long m_; int s_; asm volatile("v_cmp_eq_f32_e64 %0, %2, %3\n\ts_ff1_i32_b64 %1, %0" : "=s"(m_), "=s"(s_) : "v"(a[i]), "v"(cf)); }
            BODY8(S)
#undef S
        } else if constexpr (OP == 19) {
#define S(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : );
            BODY8(S)
#undef S
        } else if constexpr (OP == 20) {
#define S(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(cf));
            BODY8(S)
#undef S
        } else if constexpr (OP == 21) {
#define S(i) asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(a[i]) : "v"(b[i]));
            BODY8(S)
#undef S
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + (float)d[i]; for (int i = 0; i < 16; ++i) s += b[i];
    if (s == 12345.678f) out[0] = s;
}
template <int OP> int run(const char *name, int waves_per_simd) {
    float *out; CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256 * waves_per_simd;   // 256-thread blocks = 4 waves, one per SIMD
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double instr_per_simd = (double)ITER * 8 * waves_per_simd;
    printf("%-28s waves/SIMD %d  %8.3f ms  %6.3f ns per wave-instruction per SIMD\n", name, waves_per_simd, ms, ms * 1e6 / instr_per_simd);
    return 0;
}
int main() {
    for (int w : {1, 8}) {
        run<0>("v_fma_f32", w); run<1>("v_cvt_f64_f32", w); run<2>("v_cvt_f32_f64", w); run<3>("v_mul_f64", w); run<7>("v_fma_f64", w);
        run<4>("v_pk_mul_f32", w); run<5>("v_cndmask_b32", w); run<6>("v_rcp_f32", w); run<8>("dependent v_fma_f32 x8", w); run<9>("dependent cvt/mul64/cvt x8", w);
        run<10>("v_cmp+nop+v_cndmask vcc (3 instr)", w); run<11>("v_cndmask_e64 sgpr mask", w); run<19>("v_cndmask vcc, distinct src", w); run<12>("v_mov_b32", w); run<13>("v_cmp_gt_f32 vcc", w); run<14>("v_max_f32", w);
        run<15>("s_nop1 + v_max_f32_dpp (dep)", w); run<16>("v_readlane_b32", w); run<17>("v_pk_add_f32", w); run<18>("v_cmp_eq_e64 + s_ff1", w); run<20>("v_add_f32", w); run<21>("v_fma_f32 3 distinct srcs", w);
    }
    return 0;
}
