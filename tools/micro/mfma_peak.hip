// Microbenchmarks (test tooling): fp64 MFMA issue rate, and the GEMM kernel alone on large tile sets.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../gaussian_process_optimization_amd/csrc/gphip_internal.h"

// the library's launch-status slot (api_core.hip) is not linked into this stand-alone tool: report to stderr instead
void gp_note_hip(hipError_t e, const char *what, const char *file, int line) {
    if (e != hipSuccess) fprintf(stderr, "%s -> %s (%s:%d)\n", what, hipGetErrorString(e), file, line);
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(double *out, int iters) {
    double4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the same loop with operands that differ from one MFMA to the next and carry random mantissas (what a GEMM on real data feeds
// the pipe): the rate of the first loop is the issue rate at the clock the chip holds on near-constant operands, this one's is
// the ceiling of any fp64 MFMA kernel on real data
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop_rand(double *out, int iters) {
    double4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){0, 0, 0, 0};
    double a[8], b[8];
    unsigned long long h = 0x9E3779B97F4A7C15ull * (threadIdx.x + 1 + 977u * blockIdx.x);
    for (int i = 0; i < 8; ++i) {
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        a[i] = __longlong_as_double((long long)((h & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull)) - 1.5;
        h ^= h >> 31; h *= 0x94D049BB133111EBull; h ^= h >> 29;
        b[i] = __longlong_as_double((long long)((h & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull)) - 1.5;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 7]), "v"(b[(i * 3 + 1) & 7]));
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    double *out; CHK(hipMalloc(&out, 256 * 2048 * 8));
    for (int bpc : {1, 2}) {
        const int blocks = 256 * bpc, iters = 20000;
        hipLaunchKernelGGL(mfma_loop<16>, dim3(blocks), dim3(256), 0, 0, out, 100);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(mfma_loop<16>, dim3(blocks), dim3(256), 0, 0, out, iters);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        double flops = (double)blocks * 4 * iters * 16 * 2048.0;
        printf("mfma f64 16x16x4, %d waves/SIMD: %.2f TFLOP/s (%.1f ms); cycles/MFMA/SIMD at 2.4GHz: %.1f\n", bpc, flops / ms / 1e9, ms,
               ms * 1e-3 * 2.4e9 / ((double)iters * 16 * bpc));
    }
    for (int bpc : {1, 2}) {
        const int blocks = 256 * bpc, iters = 20000;
        hipLaunchKernelGGL(mfma_loop_rand<16>, dim3(blocks), dim3(256), 0, 0, out, 100);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(mfma_loop_rand<16>, dim3(blocks), dim3(256), 0, 0, out, iters);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        double flops = (double)blocks * 4 * iters * 16 * 2048.0;
        printf("mfma f64 16x16x4, RANDOM operands changing every MFMA, %d waves/SIMD: %.2f TFLOP/s (%.1f ms)\n", bpc, flops / ms / 1e9, ms);
    }
    if (getenv("MFMA_PEAK_ONLY")) return 0;
    // GEMM alone
    const long N = 16384, lda = N;
    double *A, *C;
    CHK(hipMalloc(&A, (size_t)N * N * 8)); CHK(hipMalloc(&C, (size_t)N * N * 8));
    CHK(hipMemset(C, 0, (size_t)N * N * 8));
    std::vector<double> hostA((size_t)N * 2048);
    for (size_t i = 0; i < hostA.size(); ++i) hostA[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
    for (int r = 0; r < 8; ++r) CHK(hipMemcpy(A + (size_t)r * N * 2048, hostA.data(), hostA.size() * 8, hipMemcpyHostToDevice));
    const int nt = (int)(N / 128);
    struct Cfg { int mode, K, r0, r1, c0, c1, tri; const char *name; };
    Cfg cfgs[] = {
        {1, 512, 0, nt, 0, nt, 1, "syrk tri full N, K=512"},
        {1, 1024, 0, nt, 0, nt, 1, "syrk tri full N, K=1024"},
        {1, 2048, 0, nt, 0, nt, 1, "syrk tri full N, K=2048"},
        {1, 512, 0, 79, 0, nt, 0, "rect 79x128 tiles, K=512 (cand update)"},
        {1, 1024, 0, 79, 0, nt, 0, "rect 79x128 tiles, K=1024"},
        {0, 1024, 0, 64, 0, 64, 0, "mode0 64x64 tiles K=1024"},
        {1, 128, 0, nt, 0, nt, 1, "syrk tri full N, K=128"},
        {1, 256, 0, nt, 0, nt, 1, "syrk tri full N, K=256"},
    };
    for (auto &c : cfgs) {
        TileSet ts{c.r0, c.r1, c.c0, c.c1, c.tri};
        long ntile = tileset_count(ts);
        for (int S : {0, 4, 8, 16}) {
            GemmOpt o;
            if (getenv("GEMM_WAVES8")) o.waves8 = atoi(getenv("GEMM_WAVES8"));     // the 8-wave variant the library uses from 1024 tiles on
            if (getenv("GEMM_STAGGER")) o.stagger = atoi(getenv("GEMM_STAGGER"));
            if (getenv("GEMM_SMALL")) o.small = atoi(getenv("GEMM_SMALL"));        // 64 x 64 work units
            short *dl = nullptr;
            if (S) {
                std::vector<short> l = build_tile_list(ts, S);
                if ((long)l.size() != 2 * ntile) { printf("bad list %zu vs %ld\n", l.size(), ntile); return 1; }
                CHK(hipMalloc(&dl, l.size() * 2)); CHK(hipMemcpy(dl, l.data(), l.size() * 2, hipMemcpyHostToDevice));
                o.tile_list = dl;
            }
            launch_gemm_nt(0, c.mode, C, lda, A, lda, A + 4096, lda, 1, c.K, ts, o);
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0));
            for (int r = 0; r < 3; ++r) launch_gemm_nt(0, c.mode, C, lda, A, lda, A + 4096, lda, 1, c.K, ts, o);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
            double flops = 2.0 * 128 * 128 * c.K * ntile;
            printf("%-42s S=%2d tiles %6ld: %8.3f ms  %6.2f TFLOP/s\n", c.name, S, ntile, ms, flops / ms / 1e9);
            if (dl) CHK(hipFree(dl));
        }
    }
    return 0;
}
