// Which CUs does a CU-masked stream really run on?  (hipExtStreamCreateWithCUMask on gfx950, 8 XCDs.)
// Each workgroup records (XCC_ID, HW_ID) of its first wave; the host counts distinct (xcc, se, sh, cu) tuples per mask.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
#include <tuple>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void probe(unsigned *out, int spin) {
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID, all 32 bits
        const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
    // keep the CU busy for a while so that blocks spread over everything that is enabled
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    if (a == 123.456f) out[0] = 0;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
    std::printf("CUs %d\n", ncu);
    const int blocks = 8192;
    unsigned *d;
    CK(hipMalloc(&d, blocks * 2 * sizeof(unsigned)));
    std::vector<unsigned> h(blocks * 2);
    for (int reserve : {-1, 0, 1, 8, 32, 128}) {
        hipStream_t s;
        if (reserve < 0) { CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); }
        else {
            std::vector<uint32_t> m(words, 0u);
            for (int c = reserve; c < ncu; ++c) m[c / 32] |= 1u << (c % 32);
            CK(hipExtStreamCreateWithCUMask(&s, words, m.data()));
        }
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, s, d, 20000);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, s, d, 20000);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
        std::set<std::tuple<unsigned, unsigned, unsigned, unsigned>> cus;
        std::set<unsigned> xccs;
        for (int b = 0; b < blocks; ++b) {
            const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
            cus.insert({xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15});
            xccs.insert(xcc);
        }
        std::printf("reserve %4d: distinct CUs seen %3zu on %zu XCCs, kernel %.3f ms\n", reserve, cus.size(), xccs.size(), ms);
        {
            // does block b still run on XCC b % 8?  and how are the first 512 blocks spread over CUs (2 per CU expected)?
            int match = 0;
            for (int b = 0; b < blocks; ++b) match += ((h[2 * b + 1] & 0xf) == (unsigned)(b % 8));
            std::printf("   blocks with xcc == b %% 8: %d of %d; first 16 xcc ids:", match, blocks);
            for (int b = 0; b < 16; ++b) std::printf(" %u", h[2 * b + 1] & 0xf);
            std::printf("\n");
        }
        if (reserve == 1 || reserve == 8) {
            // which tuples are missing relative to the unmasked run is printed as per-XCC counts
            unsigned cnt[16] = {0};
            for (auto &t : cus) cnt[std::get<0>(t)]++;
            std::printf("   per-XCC:");
            for (int x = 0; x < 8; ++x) std::printf(" %u", cnt[x]);
            std::printf("\n");
        }
        CK(hipStreamDestroy(s));
    }
    return 0;
}
