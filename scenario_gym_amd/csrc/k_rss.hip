// k_rss.hip -- the RSSDistances callback inside the rollout kernel, controllers in the kernel: rollout_kernel_rss<G, WV>; with
// the ego_off_road terminal condition rollout_kernel_rss_road<G, WV> (an object of its own: k_rss_road.hip).
#include "sgym_launch.hpp"

namespace sgl {
void rollout_rss(int G, int WV, bool road, dim3 grid, hipStream_t s, const RolloutArgs &a)
{
    if (WV == 8) { // 257..512 entities (no road variant at that width)
        sg::rollout_kernel_rss<64, 8><<<grid, dim3(512), 0, s>>>(SGL_ARGS(a));
        return;
    }
    if (road) {
        rollout_rss_road(G, WV, grid, s, a);
        return;
    }
#define CALL(G_, WV_) sg::rollout_kernel_rss<G_, WV_><<<grid, dim3(64 * WV_), 0, s>>>(SGL_ARGS(a))
    SGL_DISPATCH(G, WV, CALL);
#undef CALL
}
} // namespace sgl
