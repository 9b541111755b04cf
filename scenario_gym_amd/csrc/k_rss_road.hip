// k_rss_road.hip -- the RSS callback and the ego_off_road terminal condition in one kernel: rollout_kernel_rss_road<G, WV>
// (called through rollout_rss of k_rss.hip).
#include "sgym_launch.hpp"

namespace sgl {
void rollout_rss_road(int G, int WV, dim3 grid, hipStream_t s, const RolloutArgs &a)
{
#define CALL(G_, WV_) sg::rollout_kernel_rss_road<G_, WV_><<<grid, dim3(64 * WV_), 0, s>>>(SGL_ARGS(a))
    SGL_DISPATCH(G, WV, CALL);
#undef CALL
}
} // namespace sgl
