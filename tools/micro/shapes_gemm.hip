// Test tooling: throughput of the MFMA GEMM on the launch shapes the factorisation and the candidate solve use.
// build: hipcc --offload-arch=gfx950 -O3 -o shapes_gemm shapes_gemm.hip ../../gaussian_process_optimization_amd/csrc/gemm.o
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
    double *A, *C, *T;
    CHK(hipMalloc(&A, (size_t)(N + 128) * N * 8)); CHK(hipMalloc(&C, (size_t)(N + 128) * N * 8)); CHK(hipMalloc(&T, (size_t)(N + 128) * N * 8));
    CHK(hipMemset(C, 0, (size_t)N * N * 8)); CHK(hipMemset(T, 0, (size_t)N * N * 8));
    std::vector<double> hostA((size_t)N * 2048);
    for (size_t i = 0; i < hostA.size(); ++i) hostA[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
    for (int r = 0; r < 8; ++r) CHK(hipMemcpy(A + (size_t)r * N * 2048, hostA.data(), hostA.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipStream_t s0, s8;
    CHK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    std::vector<uint32_t> mask(8, 0xffffffffu);
    for (int i = 0; i < 8; ++i) mask[0] &= ~(1u << i);
    CHK(hipExtStreamCreateWithCUMask(&s8, 8, mask.data()));
    auto run = [&](const char *name, hipStream_t s, double *Cc, const double *Aa, const double *Bb, int K, TileSet ts, GemmOpt o) -> int {
        long ntile = tileset_count(ts);
        launch_gemm_nt(s, 1, Cc, lda, Aa, lda, Bb, lda, 1, K, ts, o);
        CHK(hipStreamSynchronize(s));
        CHK(hipEventRecord(e0, s));
        for (int r = 0; r < 3; ++r) launch_gemm_nt(s, 1, Cc, lda, Aa, lda, Bb, lda, 1, K, ts, o);
        CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
        printf("%-44s tiles %6ld K %4d: %8.3f ms  %6.2f TFLOP/s\n", name, ntile, K, ms, 2.0 * 128 * 128 * K * ntile / ms / 1e9);
        fflush(stdout);
        return 0;
    };
    GemmOpt o; o.stagger = 3; o.waves8 = 1;
    GemmOpt o4; o4.stagger = 3; o4.waves8 = 0;
    char nm[128];
    for (int rep = 0; rep < 2; ++rep) {
        const int K = 768, c0 = 12;
        const long off = (long)(c0 - K / 128) * 128;
        if (run("tri  C=C A=B=Apanel (syrk)", s0, C, A + off, A + off, K, TileSet{0, 129, c0, 128, 1}, o)) return 1;
        if (run("tri  C=T A=Cpanel B=Apanel", s0, T, C + off, A + off, K, TileSet{0, 129, c0, 128, 1}, o)) return 1;
        if (run("tri  C=A itself (in-place syrk as in potrf)", s0, A, A + off, A + off, K, TileSet{0, 129, c0, 128, 1}, o)) return 1;
        if (run("rect 117x59 C=C A=B=Apanel", s0, C, A + off, A + off, K, TileSet{0, 117, 12, 71, 0}, o)) return 1;
        if (run("rect 117x59 C=T A=Cpanel B=Apanel", s0, T, C + off, A + off, K, TileSet{0, 117, 12, 71, 0}, o)) return 1;
        if (run("rect 59x117 C=C A=B=Apanel", s0, C, A + off, A + off, K, TileSet{0, 59, 12, 129, 0}, o)) return 1;
    }
    return 0;
}
