// k_rss_tab.hip -- rollout_kernel_rss_tab<G>: the RSS callback with the controlled lanes on the pre-pass table, and
// rss_lines_kernel, which finishes the queued line tests of every RSS variant after a launch.
#define SG_UNIT_RSS_LINES
#include "sgym_launch.hpp"

namespace sgl {
void rollout_rss_tab(int G, dim3 grid, hipStream_t s, const sg::Params &p, double timestep, int force, const sg::TabGroups &tg)
{
#define CALL(G_) sg::rollout_kernel_rss_tab<G_><<<grid, dim3(64), 0, s>>>(p, timestep, force, tg)
    SGL_DISPATCH_G(G, CALL);
#undef CALL
}
void rollout_rss_tabq(int G, dim3 grid, hipStream_t s, const sg::Params &p, double timestep, int force, const sg::TabQueue &tq)
{
#define CALL(G_) sg::rollout_kernel_rss_tabq<G_><<<grid, dim3(64), 0, s>>>(p, timestep, force, tq)
    SGL_DISPATCH_G(G, CALL);
#undef CALL
}
int rss_tabq_waves_per_cu(int G) // (as tabq_waves_per_cu)
{
    int n = 0;
    hipError_t e = hipSuccess;
#define CALL(G_) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sg::rollout_kernel_rss_tabq<G_>, 64, 0)
    SGL_DISPATCH_G(G, CALL);
#undef CALL
    if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
void rss_lines(dim3 grid, hipStream_t s, const sg::Params &p, const sg::TabGroups &tg)
{
    sg::rss_lines_kernel<<<grid, dim3(64), 0, s>>>(p, tg);
}
} // namespace sgl
