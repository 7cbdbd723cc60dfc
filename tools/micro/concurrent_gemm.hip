// Test tooling: what two MFMA-bound launches cost when they share the chip.  Two K = 768 updates C -= A B^T on disjoint tile sets,
// (a) one after the other on one stream, (b) side by side on the library's stream kinds (CU-masked bulk stream + unmasked high-priority
// chain stream; two unmasked streams; two masked streams), with the same and with different operand panels.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../../gaussian_process_optimization_amd/csrc/gphip_internal.h"

void gp_note_hip(hipError_t e, const char *what, const char *file, int line) {
    if (e != hipSuccess) fprintf(stderr, "%s -> %s (%s:%d)\n", what, hipGetErrorString(e), file, line);
}
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static int masked_stream(hipStream_t *s, int reserve) {
    std::vector<uint32_t> mask(8, 0xffffffffu);
    for (int i = 0; i < reserve; ++i) mask[i / 32] &= ~(1u << (i % 32));
    return hipExtStreamCreateWithCUMask(s, 8, mask.data()) == hipSuccess ? 0 : 1;
}

int main(int argc, char **argv) {
    const long N = 16384, lda = N;
    const int K = argc > 1 ? atoi(argv[1]) : 768;
    double *A, *C;
    CHK(hipMalloc(&A, (size_t)N * N * 8)); CHK(hipMalloc(&C, (size_t)(N + 128) * N * 8));
    CHK(hipMemset(C, 0, (size_t)(N + 128) * N * 8));
    std::vector<double> hostA((size_t)N * 2048);
    for (size_t i = 0; i < hostA.size(); ++i) hostA[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
    for (int r = 0; r < 8; ++r) CHK(hipMemcpy(A + (size_t)r * N * 2048, hostA.data(), hostA.size() * 8, hipMemcpyHostToDevice));
    // the library's streams, in its creation order: main (low), chain (high priority), bulk (masked), inverse (low), candidates (masked)
    int lo = 0, hi = 0;
    CHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t s_main, s_chain, s_bulk, s_inv, s_pred;
    CHK(hipStreamCreateWithPriority(&s_main, hipStreamNonBlocking, lo));
    CHK(hipStreamCreateWithPriority(&s_chain, hipStreamNonBlocking, hi));
    if (masked_stream(&s_bulk, 32)) return 1;
    CHK(hipStreamCreateWithPriority(&s_inv, hipStreamNonBlocking, lo));
    if (masked_stream(&s_pred, 32)) return 1;
    struct Job { TileSet ts; long aoff; };
    const Job X{TileSet{0, 128, 0, 40, 0}, 0}, Y1{TileSet{0, 128, 40, 60, 0}, 0}, Y2{TileSet{0, 128, 40, 60, 0}, 4096};
    auto launch = [&](hipStream_t s, const Job &j) {
        GemmOpt o;
        launch_gemm_nt(s, 1, C, lda, A + j.aoff, lda, A + j.aoff, lda, 1, K, j.ts, o);
    };
    auto flops = [&](const Job &j) { return 2.0 * 128 * 128 * K * (double)tileset_count(j.ts); };
    struct Case { const char *name; hipStream_t sx, sy; const Job *y; };
    const Case cases[] = {
        {"X, Y one after the other, one unmasked stream              ", s_main, s_main, &Y2},
        {"X, Y one after the other, one masked stream (224 CUs)      ", s_bulk, s_bulk, &Y2},
        {"X masked bulk || Y unmasked high-priority, other operands  ", s_bulk, s_chain, &Y2},
        {"X masked bulk || Y unmasked high-priority, same operands   ", s_bulk, s_chain, &Y1},
        {"X unmasked    || Y unmasked (equal priority), other operands", s_main, s_inv, &Y2},
        {"X unmasked    || Y unmasked (equal priority), same operands ", s_main, s_inv, &Y1},
        {"X masked bulk || Y masked candidates, other operands       ", s_bulk, s_pred, &Y2},
        {"X unmasked    || Y unmasked high-priority, other operands  ", s_main, s_chain, &Y2},
    };
    for (int rep = 0; rep < 2; ++rep)
        for (const Case &c : cases) {
            double best = 1e30;
            for (int r = 0; r < 6; ++r) {
                CHK(hipDeviceSynchronize());
                auto t0 = std::chrono::steady_clock::now();
                launch(c.sx, X);
                launch(c.sy, *c.y);
                CHK(hipDeviceSynchronize());
                double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                if (r) best = std::min(best, ms);
            }
            printf("K %d  %s %7.3f ms  %6.2f TFLOP/s\n", K, c.name, best, (flops(X) + flops(*c.y)) / best / 1e9);
        }
    return 0;
}
