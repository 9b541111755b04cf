// k_ped_rss.hip -- the general pedestrian variants with the RSSDistances callback inside the kernel: rollout_kernel_rss_ped<G, WV>
// (called through rollout_ped of k_ped.hip; tiles of 16 lanes and more, up to four wavefronts).
#include "sgym_launch.hpp"

namespace sgl {
void rollout_ped_rss(int G, int WV, dim3 grid, hipStream_t s, const RolloutArgs &a)
{
#define CALL(G_, WV_) sg::rollout_kernel_rss_ped<(G_ < 16 ? 16 : G_), WV_><<<grid, dim3(64 * WV_), 0, s>>>(SGL_ARGS(a))
    SGL_DISPATCH(G, WV, CALL);
#undef CALL
}
} // namespace sgl
