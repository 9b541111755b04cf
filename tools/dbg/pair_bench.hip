// microbenchmark: cycles per crowd_pair evaluation of a wavefront that is alone on its SIMD
#include "../../scenario_gym_amd/csrc/sgym_device.hpp"
#include "pair_bench_pairs.hpp"
#include <cstdio>
#include <vector>
template <int N, bool INTERLEAVED>
__global__ __launch_bounds__(64, 1) void k(sg::CrowdConsts C, const double *in, double *out, unsigned long long *cyc, int iters)
{
    const int i = threadIdx.x;
    double rx[N], ry[N], ox[N], oy[N], sx[N], sy[N], ss[N], c1x[N], c1y[N], c2x[N], c2y[N], d2[N];
    bool bad[N];
    for (int u = 0; u < N; ++u) {
        rx[u] = in[i + 64 * u] + 0.5; ry[u] = in[i + 64 * (u + 1)] - 0.7; ox[u] = 0.6; oy[u] = 0.8; sx[u] = 0.01; sy[u] = 0.02; ss[u] = 0.0005;
    }
    double acc = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (INTERLEAVED) sg::crowd_pair_n<N>(C, rx, ry, ox, oy, sx, sy, ss, c1x, c1y, c2x, c2y, d2, bad);
        else
            for (int u = 0; u < N; ++u) sg::crowd_pair(C, rx[u], ry[u], ox[u], oy[u], sx[u], sy[u], ss[u], c1x[u], c1y[u], c2x[u], c2y[u], d2[u], bad[u]);
        for (int u = 0; u < N; ++u) { acc += c1x[u] + c1y[u]; rx[u] += 1e-9 * c1x[u]; ry[u] -= 1e-9 * c1y[u]; }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + i] = acc;
    if (i == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int N, bool IL>
void run(const char *name, int wgs)
{
    sg::CrowdConsts C{1.0, 1.0, 1.0, -0.17, 0.5, 0.0};
    double *in, *out; unsigned long long *cyc;
    hipMalloc(&in, 64 * 16 * 8); hipMalloc(&out, wgs * 64 * 8); hipMalloc(&cyc, wgs * 8);
    std::vector<double> h(64 * 16);
    for (size_t q = 0; q < h.size(); ++q) h[q] = 1.0 + 0.01 * (q % 97);
    hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<N, IL><<<wgs, 64>>>(C, in, out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<N, IL><<<wgs, 64>>>(C, in, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(wgs);
    hipMemcpy(c.data(), cyc, wgs * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : c) s += v;
    printf("%-28s wgs %5d: %.0f ticks per pair-eval (s_memtime), kernel %.3f ms -> %.1f ns per pair-eval -> %.2f ticks per ns\n", name, wgs, s / wgs / iters / N, ms, ms * 1e6 / iters / N, (s / wgs) / (ms * 1e6));
    hipFree(in); hipFree(out); hipFree(cyc);
}
int main()
{
    for (int wgs : {1024, 2048}) {
        run<1, false>("N=1", wgs); run<2, false>("N=2 sequential", wgs); run<4, false>("N=4 sequential", wgs);
        run<2, true>("N=2 interleaved", wgs); run<4, true>("N=4 interleaved", wgs); run<8, true>("N=8 interleaved", wgs);
    }
    return 0;
}
