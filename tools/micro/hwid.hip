// Test tooling: which (XCC, SE, SH, CU) identities workgroups report, and how many workgroups each gets.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <tuple>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void who(unsigned *out, int spin) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(100);
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; }
}
int main() {
    const int n = 4096;
    unsigned *d; CHK(hipMalloc(&d, n * 2 * 4));
    hipLaunchKernelGGL(who, dim3(n), dim3(256), 0, 0, d, 200);
    CHK(hipDeviceSynchronize());
    std::vector<unsigned> h(n * 2); CHK(hipMemcpy(h.data(), d, n * 2 * 4, hipMemcpyDeviceToHost));
    std::map<std::tuple<unsigned, unsigned, unsigned, unsigned>, int> cnt;
    for (int i = 0; i < n; ++i) {
        unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        cnt[{xcc, se, sh, cu}]++;
    }
    printf("%zu distinct (xcc,se,sh,cu)\n", cnt.size());
    unsigned lastx = 99;
    for (auto &kv : cnt) {
        auto [x, se, sh, cu] = kv.first;
        if (x != lastx) { printf("\nxcc %u:", x); lastx = x; }
        printf(" (se%u sh%u cu%u)x%d", se, sh, cu, kv.second);
    }
    printf("\nfirst 16 blocks: ");
    for (int i = 0; i < 16; ++i) printf("[b%d xcc%u se%u cu%u] ", i, h[2 * i + 1] & 0xf, (h[2 * i] >> 13) & 7, (h[2 * i] >> 8) & 0xf);
    printf("\n");
    return 0;
}
