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
} // namespace sgl
