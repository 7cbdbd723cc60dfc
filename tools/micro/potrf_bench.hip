// Diagnostic build of the diagonal-tile kernel with cycle stamps (test tooling; never shipped).
#define POTRF_STAMPS 1
#include "../../gaussian_process_optimization_amd/csrc/potrf.hip"
#include <cstdio>
#include <vector>
#include <cmath>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const int n = 128;
    std::vector<double> A(n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i * n + j] = std::exp(-0.5 * (i - j) * (i - j) / 400.0) + (i == j ? 0.01 : 0.0);
    double *dA, *dI; int *dinfo;
    CHK(hipMalloc(&dA, n * n * 8)); CHK(hipMalloc(&dI, n * n * 8)); CHK(hipMalloc(&dinfo, 16)); CHK(hipMemset(dinfo, 0, 16));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice));
        CHK(hipEventRecord(e0));
        launch_potrf_tile(0, dA, n, 0, dI, dinfo);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long st[64];
        CHK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_potrf_stamps), sizeof st));
        printf("rep %d: %.1f us total (event); stamps are s_memtime ticks (100 MHz constant clock? see below)\n", rep, ms * 1e3);
        auto d = [&](int a, int b) { return (double)(st[b] - st[a]); };
        printf("  load %.0f | potrf16(0) %.0f | loop total %.0f | inverse %.0f | store %.0f | all %.0f ticks\n", d(0, 1), d(1, 2), d(2, 24), d(24, 25), d(25, 26), d(0, 26));
        for (int p = 0; p < 7; ++p) printf("   p=%d stageA %.0f  wave0 stageB(update+potrf16+inv16) %.0f  barrier-wait %.0f\n", p, d(3 + 3 * p, 4 + 3 * p), d(4 + 3 * p, 5 + 3 * p), d(5 + 3 * p, p < 6 ? 3 + 3 * (p + 1) : 24));
    }
    int info; CHK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost)); printf("info %d\n", info);
    return 0;
}
