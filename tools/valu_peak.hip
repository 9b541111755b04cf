// valu_peak.hip -- micro-benchmark behind bench.py's secondary roof: how many vector-ALU wavefront-instructions per
// second does this MI355X issue?  ("fp64 vector ALU ... re-measure with a µbench".)
//
// Every thread runs CHAINS independent v_fma_f64 chains (full-rate fp64 on CDNA4), enough wavefronts to fill every SIMD
// (8 per SIMD); one launch is timed with HIP events.  Reported: wavefront-instructions / s over the whole chip (the unit of
// SQ_INSTS_VALU) and the fp64 rate they amount to (x 64 lanes x 2 flops).  A second kernel times ONE dependent chain per
// thread at two wavefronts per SIMD -- the shape of the rollout kernels' fp64 recurrences -- to show how far dependent
// issue falls below the peak.
//
// Built by scenario_gym_amd/csrc/Makefile into scenario_gym_amd/lib/libvalu_peak.so; C entry point valu_peak_measure().
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
constexpr int CHAINS = 8, UNROLL = 16;

__global__ __launch_bounds__(256) void fma_chains(double *out, int iters, double x, double y)
{
    double a[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) a[c] = (double)(threadIdx.x + c);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) a[c] = __builtin_fma(a[c], x, y);
    }
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += a[c];
    if (s == 12345.678) out[blockIdx.x * blockDim.x + threadIdx.x] = s; // never true: keeps the chains alive
}

// one dependent chain per thread, register budget of the rollout kernels (two wavefronts per SIMD)
__global__ __launch_bounds__(64, 2) void fma_one_chain(double *out, int iters, double x, double y)
{
    double a = (double)threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < UNROLL * CHAINS; ++u) a = __builtin_fma(a, x, y);
    }
    if (a == 12345.678) out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
} // namespace

// out[0] = peak VALU wavefront-instructions / s (whole chip), out[1] = the same as fp64 TFLOP/s,
// out[2] = wavefront-instructions / s of dependent chains at `waves_per_simd` wavefronts per SIMD (1..8), out[3] = CUs.
// Returns 0, or a HIP error code.
extern "C" int valu_peak_measure(int device, int waves_per_simd, double *out)
{
    hipDeviceProp_t prop;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return (int)e;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return (int)e;
    const int cus = prop.multiProcessorCount;
    double *d = nullptr;
    if ((e = hipMalloc((void **)&d, sizeof(double) * 256 * (size_t)cus * 8)) != hipSuccess) return (int)e;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 4000;
    float best = 1e30f, best1 = 1e30f;
    const int blocks = cus * 8; // 8 blocks of 4 wavefronts per CU = 8 wavefronts per SIMD
    const int wps = waves_per_simd < 1 ? 1 : (waves_per_simd > 8 ? 8 : waves_per_simd);
    const int blocks1 = cus * 4 * wps; // one-wavefront blocks
    for (int rep = 0; rep < 4; ++rep) {
        float ms = 0.f;
        (void)hipEventRecord(e0, 0);
        fma_chains<<<dim3(blocks), dim3(256), 0, 0>>>(d, iters, 0.999999, 1e-9);
        (void)hipEventRecord(e1, 0);
        if ((e = hipEventSynchronize(e1)) != hipSuccess) break;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
        (void)hipEventRecord(e0, 0);
        fma_one_chain<<<dim3(blocks1), dim3(64), 0, 0>>>(d, iters, 0.999999, 1e-9);
        (void)hipEventRecord(e1, 0);
        if ((e = hipEventSynchronize(e1)) != hipSuccess) break;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best1) best1 = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(d);
    if (e != hipSuccess) return (int)e;
    const double per_wave = (double)iters * UNROLL * CHAINS;
    out[0] = per_wave * blocks * 4 / (best * 1e-3);
    out[1] = out[0] * 64 * 2 / 1e12;
    out[2] = per_wave * blocks1 / (best1 * 1e-3);
    out[3] = cus;
    return 0;
}
