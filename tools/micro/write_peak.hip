// Write-only bandwidth of this card for the K-build's store pattern (test tooling; never shipped).
//   hipcc --offload-arch=gfx950 -O3 write_peak.hip -o write_peak
// (a) hipMemsetAsync of 2.15 GB; (b) a kernel storing 16 bytes per lane, 128 x 128 tiles of a 16384 x 16384 row-major matrix, lower
// tiles only (the K-build's pattern: 1 KiB contiguous per wave and row), values from a trivial computation; (c) the same with
// ~40 dependent f64 FMAs per element in front of the store (the K-build's arithmetic, no exp).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double2_t __attribute__((ext_vector_type(2)));
template <int WORK>
__global__ __launch_bounds__(256) void tile_store(double *A, long lda, int nt) {
    const long t = blockIdx.x;
    int tm = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((long)tm * (tm + 1) / 2 > t) --tm;
    while ((long)(tm + 1) * (tm + 2) / 2 <= t) ++tm;
    const int tn = (int)(t - (long)tm * (tm + 1) / 2);
    const int cx = (threadIdx.x & 63) * 2, ry = threadIdx.x >> 6;
    for (int q = 0; q < 32; ++q) {
        const long gr = (long)tm * 128 + ry + 4 * q, gc = (long)tn * 128 + cx;
        double2_t v = {(double)gr, (double)gc};
        for (int i = 0; i < WORK; ++i) { v[0] = fma(v[0], 0.999, 1e-3); v[1] = fma(v[1], 0.999, 1e-3); }
        *(double2_t *)(A + gr * lda + gc) = v;
    }
}
int main() {
    const long N = 16384; const int nt = 128;
    double *A; hipMalloc(&A, N * N * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0); hipMemsetAsync(A, 0, N * N * 8); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("memset 2.15 GB: %.3f ms = %.2f TB/s\n", ms, N * N * 8 / ms / 1e9);
        const long nblk = (long)nt * (nt + 1) / 2;
        hipEventRecord(e0); hipLaunchKernelGGL(tile_store<0>, dim3(nblk), dim3(256), 0, 0, A, N, nt); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("lower tiles, store only: %.3f ms = %.2f TB/s\n", ms, nblk * 131072.0 / ms / 1e9);
        hipEventRecord(e0); hipLaunchKernelGGL(tile_store<20>, dim3(nblk), dim3(256), 0, 0, A, N, nt); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("lower tiles, 40 FMAs per element + store: %.3f ms = %.2f TB/s\n", ms, nblk * 131072.0 / ms / 1e9);
        hipEventRecord(e0); hipLaunchKernelGGL(tile_store<40>, dim3(nblk), dim3(256), 0, 0, A, N, nt); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("lower tiles, 80 FMAs per element + store: %.3f ms = %.2f TB/s\n", ms, nblk * 131072.0 / ms / 1e9);
    }
    return 0;
}
