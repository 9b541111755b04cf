// Does a fp64 VALU instruction of a wavefront with fewer active lanes issue faster on gfx950?  (16 fp64 lanes per SIMD per clock:
// a full wavefront takes 4 passes.)  One wavefront per SIMD-ish, chains of dependent / independent v_fma_f64 with 64, 32, 16 and
// 8 active lanes; prints cycles per instruction.   hipcc --offload-arch=gfx950 -O3 -o /tmp/dp_lanes tools/dbg/dp_lanes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain(double *out, long long *cyc, int active, int indep, int iters)
{
    const int lane = threadIdx.x & 63;
    double a0 = 1.0 + lane * 1e-9, a1 = 1.1, a2 = 1.2, a3 = 1.3, b = 1.0000001, c = 1e-9;
    long long t0 = 0, t1 = 0;
    if (lane < active) {
        t0 = clock64();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                a0 = __builtin_fma(a0, b, c);
                if (indep) { a1 = __builtin_fma(a1, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c); }
            }
        }
        t1 = clock64();
        out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3;
        if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    }
}
int main()
{
    double *out; long long *cyc;
    hipMalloc(&out, 1024 * 64 * 8); hipMalloc(&cyc, 1024 * 8);
    const int iters = 4096;
    for (int indep = 0; indep < 2; ++indep)
        for (int active : {64, 32, 16, 8}) {
            for (int blocks : {1, 1024, 3072}) {
                hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, out, cyc, active, indep, iters);
                hipDeviceSynchronize();
                long long h[4]; hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
                const double per = (double)h[0] / ((double)iters * 16 * (indep ? 4 : 1));
                printf("%s chain, %2d active lanes, %4d wavefronts: %.2f shader cycles per v_fma_f64\n", indep ? "4 independent" : "dependent", active, blocks, per);
            }
        }
    return 0;
}
