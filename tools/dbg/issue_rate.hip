// What one wavefront pays per instruction on gfx950, in NANOSECONDS (HIP events over a long launch: no clock to calibrate):
// fp64 fma dependent / 8 independent chains, fp32 fma, v_cndmask, s_mov-like scalar work, at 1, 2 and 3 wavefronts per SIMD
// (256 blocks of 256 / 512 / 768 threads: one block per CU, its wavefronts spread over the four SIMDs).
//   hipcc --offload-arch=gfx950 -O3 -w -o /tmp/issue_rate tools/dbg/issue_rate.hip && /tmp/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(double *out, int iters)
{
    double a[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = 1.0 + threadIdx.x * 1e-9 + i; f[i] = 1.0f + i; }
    const double b = 1.0000001, c = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) { a[0] = __builtin_fma(a[0], b, c); }                                    // dependent fp64 fma
            if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = __builtin_fma(a[i], b, c);                        // 8 independent fp64 fma
            }
            if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) f[i] = __builtin_fmaf(f[i], 1.0000001f, 1e-9f);          // 8 independent fp32 fma
            }
            if (MODE == 3) {                                                                        // 8 independent fp64 mul + add (unfused)
#pragma unroll
                for (int i = 0; i < 8; ++i) { a[i] = a[i] * b; asm volatile("" : "+v"(a[i])); a[i] = a[i] + c; }
            }
            if (MODE == 4) { f[0] = __builtin_fmaf(f[0], 1.0000001f, 1e-9f); }                      // dependent fp32 fma
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
static void run(const char *name, int per_iter, double *out)
{
    for (int threads : {256, 512, 768}) {
        const int iters = 1 << 17;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters / 8);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)iters * 8 * per_iter;
        printf("%-34s %d wavefront(s) per SIMD: %6.2f ns per instruction per wavefront (%5.2f ns per instruction per SIMD)\n", name, threads / 256,
               ms * 1e6 / n, ms * 1e6 / n / (threads / 256));
    }
}
int main()
{
    double *out;
    hipMalloc(&out, 256 * 768 * 8);
    run<0>("fp64 fma, dependent", 1, out);
    run<1>("fp64 fma, 8 independent", 8, out);
    run<3>("fp64 mul + add, 8 independent", 16, out);
    run<4>("fp32 fma, dependent", 1, out);
    run<2>("fp32 fma, 8 independent", 8, out);
    return 0;
}
