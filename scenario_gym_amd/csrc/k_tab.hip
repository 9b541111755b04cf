// k_tab.hip -- the headline kernels (BASELINE configs 3 / 4): rollout_kernel_tab<G> and rollout_kernel_tab_planar<G>,
// controlled lanes replayed from the pre-pass table, three wavefronts per SIMD.
#include "sgym_launch.hpp"

namespace sgl {
void rollout_tab(int G, bool planar, dim3 grid, hipStream_t s, const sg::Params &p, double timestep, int force, const sg::TabGroups &tg)
{
#define CALL(G_)                                                                                                                     \
    if (planar) sg::rollout_kernel_tab_planar<G_><<<grid, dim3(64), 0, s>>>(p, timestep, force, tg);                                 \
    else sg::rollout_kernel_tab<G_><<<grid, dim3(64), 0, s>>>(p, timestep, force, tg)
    SGL_DISPATCH_G(G, CALL);
#undef CALL
}
} // namespace sgl
