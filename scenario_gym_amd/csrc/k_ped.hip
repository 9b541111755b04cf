// k_ped.hip -- the general pedestrian variants rollout_kernel<G, WV, true, false> (tiles narrower than 16 lanes run as
// 16-lane tiles); the variants with the RSS callback are an object of their own (k_ped_rss.hip: they compile as long again).
#include "sgym_launch.hpp"

namespace sgl {
void rollout_ped(int G, int WV, bool rss, dim3 grid, hipStream_t s, const RolloutArgs &a)
{
    if (WV == 8) { // 257..512 entities: the general pedestrian variant on eight wavefronts (no RSS callback at that width)
        sg::rollout_kernel<64, 8, true, false><<<grid, dim3(512), 0, s>>>(SGL_ARGS(a));
        return;
    }
    if (WV == 1 && G < 16) G = 16;
    if (rss) {
        rollout_ped_rss(G, WV, grid, s, a);
        return;
    }
#define CALL(G_, WV_) sg::rollout_kernel<(G_ < 16 ? 16 : G_), WV_, true, false><<<grid, dim3(64 * WV_), 0, s>>>(SGL_ARGS(a))
    SGL_DISPATCH(G, WV, CALL);
#undef CALL
}
} // namespace sgl
