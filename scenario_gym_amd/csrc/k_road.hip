// k_road.hip -- rollout_kernel_road<G, WV>: the ego_off_road terminal condition compiled in.
#include "sgym_launch.hpp"

namespace sgl {
void rollout_road(int G, int WV, dim3 grid, hipStream_t s, const RolloutArgs &a)
{
    if (WV == 8) { // 257..512 entities
        sg::rollout_kernel_road<64, 8><<<grid, dim3(512), 0, s>>>(SGL_ARGS(a));
        return;
    }
#define CALL(G_, WV_) sg::rollout_kernel_road<G_, WV_><<<grid, dim3(64 * WV_), 0, s>>>(SGL_ARGS(a))
    SGL_DISPATCH(G, WV, CALL);
#undef CALL
}
} // namespace sgl
