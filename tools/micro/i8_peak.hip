// Test tooling: issue rate of v_mfma_i32_32x32x32_i8 (operands in registers, random bytes, every CU), with 1 or 2 waves
// per SIMD and with the accumulators re-used back to back (1 chain) or rotated over 4 / 8 independent accumulators.
// What the int8 residue GEMM of rns.hip (1.5-1.7 Pop/s) has to be read against.
// build: hipcc --offload-arch=gfx950 -O3 -o i8_peak i8_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int int4_t __attribute__((ext_vector_type(4)));
typedef int int16v_t __attribute__((ext_vector_type(16)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void loop(int *out, int iters, unsigned seed) {
    int16v_t acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    unsigned x = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    int4_t a, b;
    for (int r = 0; r < 4; ++r) {
        x = x * 1664525u + 1013904223u; a[r] = (int)x;
        x = x * 1664525u + 1013904223u; b[r] = (int)x;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
    }
    int s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
int run(int *out, int bpc) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int blocks = 256 * bpc, iters = 40000 / NACC * 8;
    hipLaunchKernelGGL(loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, 100, 1u);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 7u);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double n_mfma_per_simd = (double)iters * NACC * bpc;
    const double ops = (double)blocks * 4 * iters * NACC * 65536.0;
    printf("i8 32x32x32, %d accumulators, %d wave(s)/SIMD: %.2f Pop/s (%.1f ms); %.1f cycles per MFMA per SIMD if the clock were 2.4 GHz\n",
           NACC, bpc, ops / ms / 1e12, ms, ms * 1e-3 * 2.4e9 / n_mfma_per_simd);
    return 0;
}

int main() {
    int *out; CHK(hipMalloc(&out, 256 * 2 * 256 * 4));
    for (int bpc : {1, 2}) {
        if (run<1>(out, bpc)) return 1;
        if (run<4>(out, bpc)) return 1;
        if (run<8>(out, bpc)) return 1;
    }
    return 0;
}
