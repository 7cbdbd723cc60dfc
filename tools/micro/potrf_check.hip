// Stand-alone check + cycle stamps of the diagonal-tile kernel (test tooling; never shipped).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -DPOTRF_SRC='"<path to potrf.hip>"' potrf_check.hip -o potrf_check
// Factors several 128x128 tiles (well / ill conditioned, identity padding, a non-positive pivot, a subnormal pivot), compares L and L^-1 with a
// long-double host Cholesky / substitution, and prints the kernel's own cycle stamps.
#define POTRF_STAMPS 1
#include POTRF_SRC
#include <cmath>
#include <cstdio>
#include <vector>

// the library's launch-status slot (api_core.hip) is not linked into this stand-alone tool: report to stderr instead
void gp_note_hip(hipError_t e, const char *what, const char *file, int line) {
    if (e != hipSuccess) fprintf(stderr, "%s -> %s (%s:%d)\n", what, hipGetErrorString(e), file, line);
}
#ifndef POTRF_NSTAMPS
#define POTRF_NSTAMPS 32
#endif
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static int host_chol(const std::vector<double> &A, int n, std::vector<long double> &L) {
    L.assign((size_t)n * n, 0.0L);
    for (int j = 0; j < n; ++j) {
        long double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
        if (!(d > 0)) return j + 1;
        L[j * n + j] = sqrtl(d);
        for (int i = j + 1; i < n; ++i) {
            long double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = s / L[j * n + j];
        }
    }
    return 0;
}
static void host_inv(const std::vector<long double> &L, int n, std::vector<long double> &I) {
    I.assign((size_t)n * n, 0.0L);
    for (int c = 0; c < n; ++c)
        for (int i = c; i < n; ++i) {
            long double s = (i == c) ? 1.0L : 0.0L;
            for (int k = c; k < i; ++k) s -= L[i * n + k] * I[k * n + c];
            I[i * n + c] = s / L[i * n + i];
        }
}

int main() {
    const int n = 128, lda = 384, t = 1;      // the tile sits at (1, 1) of a 3 x 3 tile matrix: lda and the tile offset are exercised
    double *dA, *dI; int *dinfo;
    CHK(hipMalloc(&dA, (size_t)lda * lda * 8)); CHK(hipMalloc(&dI, 3 * n * n * 8)); CHK(hipMalloc(&dinfo, 16));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    int bad_total = 0;
    for (int tc = 0; tc < 7; ++tc) {
        std::vector<double> A((size_t)n * n);
        int expect_info = 0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double v;
                if (tc == 0) v = std::exp(-0.5 * (i - j) * (i - j) / 400.0) + (i == j ? 0.01 : 0.0);
                else if (tc == 1) v = std::exp(-0.5 * (i - j) * (i - j) / 4000.0) + (i == j ? 1e-6 : 0.0);     // cond ~ 1e8
                else if (tc == 2) { const int a = i < 100 ? i : -1, b = j < 100 ? j : -1;                        // identity padding
                    v = (a >= 0 && b >= 0) ? 3.0 * std::exp(-std::fabs((double)(a - b)) / 7.0) + (a == b ? 0.5 : 0) : (i == j ? 1.0 : 0.0); }
                else if (tc == 3) v = (i == j ? 2.0 + 0.01 * i : 1.0 / (1.0 + std::abs(i - j)) * ((i + j) % 3 == 0 ? -0.5 : 0.4) / 8.0);
                else if (tc == 4) v = std::exp(-0.5 * (i - j) * (i - j) / 400.0) + (i == j ? (i == 77 ? -5.0 : 0.01) : 0.0);   // pivot 78 fails
                else if (tc == 6) v = (i == j) ? (i == 37 ? 1e-310 : 1.0) : 0.0;   // a positive SUBNORMAL pivot: its reciprocal overflows -> reported like a non-positive one (column 38)
                else v = std::exp(-0.5 * (i - j) * (i - j) / 50.0) * (1.0 + 0.3 * std::sin(i * 0.37) * std::sin(j * 0.37)) + (i == j ? 0.05 : 0);
                A[i * n + j] = v;
            }
        std::vector<long double> L, Iv;
        const int hinfo = host_chol(A, n, L);
        if (hinfo) expect_info = t * n + hinfo;
        if (tc == 6) expect_info = t * n + 38;
        std::vector<double> big((size_t)lda * lda, 777.0);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) big[(size_t)(t * n + i) * lda + t * n + j] = A[i * n + j];
        CHK(hipMemcpy(dA, big.data(), big.size() * 8, hipMemcpyHostToDevice));
        CHK(hipMemset(dI, 0, 3 * n * n * 8));
        CHK(hipMemset(dinfo, 0, 16));
        CHK(hipEventRecord(e0));
        launch_potrf_tile(0, dA, lda, t, dI, dinfo);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        int info; CHK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost));
        std::vector<double> out((size_t)lda * lda), inv((size_t)3 * n * n);
        CHK(hipMemcpy(out.data(), dA, out.size() * 8, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(inv.data(), dI, inv.size() * 8, hipMemcpyDeviceToHost));
        if (hinfo || tc == 6) { printf("case %d: host info %d, device info %d (expected %d) %s\n", tc, hinfo, info, expect_info, info == expect_info ? "OK" : "MISMATCH"); bad_total += info != expect_info; continue; }
        host_inv(L, n, Iv);
        double eL = 0, eI = 0, mL = 0, mI = 0, eOut = 0, eUp = 0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                const double l = out[(size_t)(t * n + i) * lda + t * n + j], iv = inv[(size_t)t * n * n + i * n + j];
                if (j <= i) { eL = std::fmax(eL, std::fabs(l - (double)L[i * n + j])); mL = std::fmax(mL, std::fabs((double)L[i * n + j]));
                              eI = std::fmax(eI, std::fabs(iv - (double)Iv[i * n + j])); mI = std::fmax(mI, std::fabs((double)Iv[i * n + j])); }
                else eUp = std::fmax(eUp, std::fabs(iv));
            }
        // nothing outside the tile may change; other tiles of the inverse workspace stay zero
        for (int i = 0; i < lda; ++i) for (int j = 0; j < lda; ++j) { const bool in = i >= t * n && i < (t + 1) * n && j >= t * n && j < (t + 1) * n; if (!in) eOut = std::fmax(eOut, std::fabs(out[(size_t)i * lda + j] - 777.0)); }
        for (int q = 0; q < 3 * n * n; ++q) if (q / (n * n) != t) eOut = std::fmax(eOut, std::fabs(inv[q]));
        // cond-aware yardstick: || I - L Iv ||
        double res = 0;
        for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { long double s = 0; for (int k = j; k <= i; ++k) s += (long double)out[(size_t)(t * n + i) * lda + t * n + k] * inv[(size_t)t * n * n + k * n + j]; res = std::fmax(res, std::fabs((double)(s - (i == j ? 1.0L : 0.0L)))); }
        const bool ok = info == 0 && eL <= 1e-10 * mL * (tc == 1 ? 1e4 : 1) && eI <= 1e-9 * mI * (tc == 1 ? 1e5 : 1) && eUp == 0 && eOut == 0 && res < 1e-9;
        printf("case %d: %.1f us  info %d  |dL| %.2e (max %.2e)  |dInv| %.2e (max %.2e)  |L Inv - I| %.2e  upper %.1e outside %.1e  %s\n", tc, ms * 1e3, info, eL, mL, eI, mI, res, eUp, eOut, ok ? "OK" : "FAIL");
        bad_total += !ok;
    }
    // timing + stamps on case 0 again
    {
        std::vector<double> big((size_t)lda * lda, 0.0);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) big[(size_t)(t * n + i) * lda + t * n + j] = std::exp(-0.5 * (i - j) * (i - j) / 400.0) + (i == j ? 0.01 : 0.0);
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipMemcpy(dA, big.data(), big.size() * 8, hipMemcpyHostToDevice));
            CHK(hipEventRecord(e0));
            launch_potrf_tile(0, dA, lda, t, dI, dinfo);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long st[64];
            CHK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_potrf_stamps), sizeof st));
            printf("rep %d: %.1f us (event).  stamps:", rep, ms * 1e3);
            for (int i = 0; i < POTRF_NSTAMPS; ++i) printf(" %llu", st[i] - st[0]);
            printf("\n");
        }
    }
    printf("%s\n", bad_total ? "POTRF CHECK FAILED" : "POTRF CHECK PASSED");
    return bad_total ? 2 : 0;
}
