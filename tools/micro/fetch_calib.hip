// Test tooling: calibrates rocprofv3's FETCH_SIZE on gfx950 for the two read patterns of gemm_nt_kernel.
//   stream16 : 16 bytes per lane, 1 KiB contiguous per wave instruction (the operand-panel staging loads)
//   ctile8   : buffer_load_b64, 8 bytes per lane; a wave instruction touches 4 rows x 128 contiguous bytes with the
//              row stride of the C matrix (the C-tile prologue of the C -= A B^T kernel)
// Each kernel reads a KNOWN number of distinct bytes exactly once (buffers far larger than the 256 MiB Infinity
// Cache).  Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and divide: bytes / (FETCH_SIZE KB * 1024) is the
// factor to apply to that pattern's share of FETCH_SIZE (tools/pmc_traffic.py).
// build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double double2_t __attribute__((ext_vector_type(2)));
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void stream16(const double2_t *src, long n16, double *sink) {
    double acc = 0.0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
        const double2_t v = src[i];
        acc += v[0] + v[1];
    }
    if (acc == 1.2345e300) sink[0] = acc;
}

// one workgroup reads one 128 x 128 tile of a row-major matrix with leading dimension ld, exactly as the 8-wave GEMM's
// prologue does: wave (wm, wn) owns rows wm*64.., columns wn*32..; lane (li = lane & 15, lg = lane >> 4) reads
// C[row = m*16 + 4r + lg][col = n*16 + li]
__global__ __launch_bounds__(512) void ctile8(const double *C, long ld, int tiles_per_row, double *sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / 4, wn = wave % 4, li = lane & 15, lg = lane >> 4;
    const long ti = blockIdx.x / tiles_per_row, tc = blockIdx.x % tiles_per_row;
    const double *Cw = C + (ti * 128 + wm * 64) * ld + tc * 128 + wn * 32;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)Cw, 0, 0x7fffffff, 0x00020000);
    const unsigned cbyte = (unsigned)(lg * (int)ld + li) * 8u, crow = (unsigned)ld * 8u;
    double acc = 0.0;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const v2u_t v = __builtin_amdgcn_raw_buffer_load_b64(rs, cbyte + n * 128, (unsigned)(m * 16 + 4 * r) * crow, 0);
                acc += __hiloint2double((int)v[1], (int)v[0]);
            }
    if (acc == 1.2345e300) sink[0] = acc;
}

int main() {
    const long N = 16384;                      // 2 GiB matrix
    double *A, *sink;
    CHK(hipMalloc(&A, (size_t)N * N * 8));
    CHK(hipMalloc(&sink, 64));
    CHK(hipMemset(A, 0, (size_t)N * N * 8));
    CHK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stream16, dim3(8192), dim3(256), 0, 0, (const double2_t *)A, N * N / 2, sink);
        hipLaunchKernelGGL(ctile8, dim3(128 * 128), dim3(512), 0, 0, A, N, 128, sink);
    }
    CHK(hipDeviceSynchronize());
    printf("stream16 reads %ld bytes per launch; ctile8 reads %ld bytes per launch\n", N * N * 8, N * N * 8);
    return 0;
}
