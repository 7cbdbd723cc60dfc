// Test tooling: does hipExtStreamCreateWithCUMask confine a stream's workgroups, and how are CU bits laid out?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <set>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void where(unsigned *out, int spin) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hwid; out[2 * blockIdx.x + 1] = xcc; }
}

int main() {
    hipDeviceProp_t pr; CHK(hipGetDeviceProperties(&pr, 0));
    printf("CUs %d\n", pr.multiProcessorCount);
    unsigned *out; CHK(hipMalloc(&out, 8 * 4096));
    std::vector<unsigned> h(2 * 4096);
    auto run = [&](hipStream_t s, const char *name) -> int {
        hipLaunchKernelGGL(where, dim3(2048), dim3(256), 0, s, out, 200000);
        CHK(hipStreamSynchronize(s));
        CHK(hipMemcpy(h.data(), out, 8 * 2048, hipMemcpyDeviceToHost));
        std::set<unsigned> cus; std::set<unsigned> xccs;
        for (int i = 0; i < 2048; ++i) {
            unsigned hw = h[2 * i], x = h[2 * i + 1] & 0xf;
            unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
            cus.insert((x << 16) | (se << 8) | (sh << 4) | cu); xccs.insert(x);
        }
        printf("%s: distinct CUs used %zu, XCCs %zu\n", name, cus.size(), xccs.size());
        int per[8] = {0};
        for (unsigned c : cus) per[c >> 16]++;
        printf("   CUs per XCC:"); for (int x = 0; x < 8; ++x) printf(" %d", per[x]); printf("\n");
        return 0;
    };
    hipStream_t s0; CHK(hipStreamCreate(&s0));
    if (run(s0, "unmasked")) return 1;
    for (int variant = 0; variant < 5; ++variant) {
        std::vector<uint32_t> mask(8, 0xffffffffu);
        if (variant == 0) mask[0] = 0xffff0000u;            // drop CUs 0..15
        if (variant == 1) for (int w = 0; w < 8; ++w) mask[w] = 0xfffffffcu;  // drop 2 bits per word
        if (variant == 2) { for (int w = 0; w < 8; ++w) mask[w] = 0; mask[0] = 0xffffu; }  // only CUs 0..15
        if (variant == 3) mask[0] = 0xffffff00u;  // drop bits 0..7
        if (variant == 4) { for (int w = 0; w < 8; ++w) mask[w] = 0; mask[0] = 0xffu; }  // only bits 0..7
        hipStream_t sm;
        hipError_t e = hipExtStreamCreateWithCUMask(&sm, 8, mask.data());
        if (e != hipSuccess) { printf("variant %d: create failed: %s\n", variant, hipGetErrorString(e)); continue; }
        char nm[64]; snprintf(nm, sizeof nm, "masked variant %d", variant);
        if (run(sm, nm)) return 1;
        CHK(hipStreamDestroy(sm));
    }
    return 0;
}
