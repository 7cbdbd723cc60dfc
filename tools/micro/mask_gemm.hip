// Test tooling: GEMM throughput on CU-masked streams vs an unmasked stream.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include "../../gaussian_process_optimization_amd/csrc/gphip_internal.h"

// the library's launch-status slot (api_core.hip) is not linked into this stand-alone tool: report to stderr instead
void gp_note_hip(hipError_t e, const char *what, const char *file, int line) {
    if (e != hipSuccess) fprintf(stderr, "%s -> %s (%s:%d)\n", what, hipGetErrorString(e), file, line);
}
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const long N = 16384, lda = N;
    double *A, *C;
    CHK(hipMalloc(&A, (size_t)N * N * 8)); CHK(hipMalloc(&C, (size_t)N * N * 8));
    CHK(hipMemset(C, 0, (size_t)N * N * 8));
    std::vector<double> hostA((size_t)N * 2048);
    for (size_t i = 0; i < hostA.size(); ++i) hostA[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
    for (int r = 0; r < 8; ++r) CHK(hipMemcpy(A + (size_t)r * N * 2048, hostA.data(), hostA.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int nt = 128;
    for (int w8 : {0, 1, 0, 1}) for (int stg : {0, 3}) for (int reserve : {0}) {
        hipStream_t s;
        std::vector<uint32_t> mask(8, 0xffffffffu);
        for (int i = 0; i < reserve; ++i) mask[i / 32] &= ~(1u << (i % 32));
        if (reserve) CHK(hipExtStreamCreateWithCUMask(&s, 8, mask.data())); else CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        TileSet ts{0, 79, 0, nt, 0};
        long ntile = tileset_count(ts);
        GemmOpt o; o.stagger = stg; o.waves8 = w8;
        for (int w = 0; w < 3; ++w) launch_gemm_nt(s, 1, C, lda, A, lda, A + 4096, lda, 1, 1024, ts, o);
        CHK(hipStreamSynchronize(s));
        CHK(hipEventRecord(e0, s));
        for (int r = 0; r < 3; ++r) launch_gemm_nt(s, 1, C, lda, A, lda, A + 4096, lda, 1, 1024, ts, o);
        CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
        printf("waves8 %d stagger %d reserve %2d CUs: %8.3f ms  %6.2f TFLOP/s (x%.3f of CUs)\n", w8, stg, reserve, ms, 2.0 * 128 * 128 * 1024 * ntile / ms / 1e9, (256.0 - reserve) / 256.0);
    }
    return 0;
}
