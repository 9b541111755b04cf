// k_tabq.hip -- the table path as ONE persistent launch (sgym_queue.hpp): rollout_kernel_tabq<G> / rollout_kernel_tabq_planar<G>,
// pre-pass and rollout roles in one grid, work items (chunk, block) from a device-side counter.
#include "sgym_launch.hpp"

namespace sgl {
void rollout_tabq(int G, bool planar, dim3 grid, hipStream_t s, const sg::Params &p, double timestep, int force, const sg::TabQueue &tq)
{
#define CALL(G_)                                                                                                                     \
    if (planar) sg::rollout_kernel_tabq_planar<G_><<<grid, dim3(64), 0, s>>>(p, timestep, force, tq);                                \
    else sg::rollout_kernel_tabq<G_><<<grid, dim3(64), 0, s>>>(p, timestep, force, tq)
    SGL_DISPATCH_G(G, CALL);
#undef CALL
}
// wavefronts (= 64-thread workgroups) of the persistent kernel one compute unit holds at once, as the runtime sees it for THIS
// device and process (a CU mask, a partition, registers or LDS the build changed); 0: the query failed
int tabq_waves_per_cu(int G, bool planar)
{
    int n = 0;
    hipError_t e = hipSuccess;
#define CALL(G_)                                                                                                                     \
    e = planar ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sg::rollout_kernel_tabq_planar<G_>, 64, 0)                         \
               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sg::rollout_kernel_tabq<G_>, 64, 0)
    SGL_DISPATCH_G(G, CALL);
#undef CALL
    if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
} // namespace sgl
