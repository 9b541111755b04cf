// Counter calibration for THIS library's access shape (MI355X_MICROARCH.md, "HBM": widths other than 16 B/lane must be
// calibrated on a known byte count): every wavefront streams whole 512-byte rows, 8 B per lane -- what the rollout kernels'
// state rows, grid rows and dnorm rows are.  Three kernels over a buffer far larger than the 256 MB Infinity Cache:
//   calib_read   reads n bytes (sum kept live), writes 8 B per wavefront
//   calib_write  writes n bytes, reads nothing
//   calib_copy   reads n and writes n
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes; tools/hbm_calib.sh) and divide.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void calib_read(const double *__restrict__ in, double *__restrict__ out, size_t rows_per_wave)
{
    const size_t wave = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const double *p = in + wave * rows_per_wave * 64 + (threadIdx.x & 63);
    double s = 0.0;
    for (size_t r = 0; r < rows_per_wave; ++r) s += p[r * 64];
    if (s == 1.2345e300) out[wave] = s; // never true: keeps the loads
}

__global__ void calib_write(double *__restrict__ out, size_t rows_per_wave, double v)
{
    const size_t wave = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    double *p = out + wave * rows_per_wave * 64 + (threadIdx.x & 63);
    for (size_t r = 0; r < rows_per_wave; ++r) p[r * 64] = v + (double)r;
}

__global__ void calib_copy(const double *__restrict__ in, double *__restrict__ out, size_t rows_per_wave)
{
    const size_t wave = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const size_t o = wave * rows_per_wave * 64 + (threadIdx.x & 63);
    for (size_t r = 0; r < rows_per_wave; ++r) out[o + r * 64] = in[o + r * 64];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const size_t gib = argc > 1 ? (size_t)atol(argv[1]) : 2; // bytes streamed per kernel = gib GiB
    const size_t waves = 256 * 4 * 8;                       // 8 wavefronts per SIMD
    const size_t rows_per_wave = (gib << 30) / 512 / waves;
    const size_t n = waves * rows_per_wave * 64;
    double *a, *b;
    CK(hipMalloc(&a, n * 8));
    CK(hipMalloc(&b, n * 8));
    CK(hipMemset(a, 0, n * 8));
    CK(hipMemset(b, 0, n * 8));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms[3];
    for (int k = 0; k < 3; ++k) {
        CK(hipEventRecord(e0));
        if (k == 0) calib_read<<<waves / 4, 256>>>(a, b, rows_per_wave);
        if (k == 1) calib_write<<<waves / 4, 256>>>(b, rows_per_wave, 1.0);
        if (k == 2) calib_copy<<<waves / 4, 256>>>(a, b, rows_per_wave);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[k], e0, e1));
    }
    printf("{\"bytes_per_kernel\": %zu, \"read_ms\": %.4f, \"write_ms\": %.4f, \"copy_ms\": %.4f, "
           "\"read_GBs\": %.1f, \"write_GBs\": %.1f, \"copy_GBs\": %.1f}\n",
           n * 8, ms[0], ms[1], ms[2], n * 8 / ms[0] / 1e6, n * 8 / ms[1] / 1e6, 2.0 * n * 8 / ms[2] / 1e6);
    return 0;
}
