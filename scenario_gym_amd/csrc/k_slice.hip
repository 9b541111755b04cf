// k_slice.hip -- the slices of a time-sliced rollout: rollout_kernel_slice<G> / rollout_kernel_slice_tab<G>.
#include "sgym_launch.hpp"

namespace sgl {
void rollout_slice(int G, dim3 grid, hipStream_t s, const sg::Params &p, double timestep, const sg::SliceArgs &sa, const double *tab)
{
#define CALL(G_)                                                                                                                     \
    if (tab) sg::rollout_kernel_slice_tab<G_><<<grid, dim3(64), 0, s>>>(p, timestep, sa, tab);                                       \
    else sg::rollout_kernel_slice<G_><<<grid, dim3(64), 0, s>>>(p, timestep, sa)
    SGL_DISPATCH_G(G, CALL);
#undef CALL
}
} // namespace sgl
