// k_plain.hip -- rollout_kernel<G, WV, false, TAB>: replay / vehicle / PID lanes with the controllers in the kernel (TAB =
// false) or batches without controlled lanes on the table path (TAB = true); rollout_kernel<64, 8, false, false>: 257..512 entities.
#include "sgym_launch.hpp"

namespace sgl {
void rollout_plain(int G, int WV, bool tab, dim3 grid, hipStream_t s, const RolloutArgs &a)
{
    if (WV == 8) {
        sg::rollout_kernel<64, 8, false, false><<<grid, dim3(512), 0, s>>>(SGL_ARGS(a));
        return;
    }
#define CALL(G_, WV_)                                                                                                                \
    if (tab) sg::rollout_kernel<G_, WV_, false, true><<<grid, dim3(64 * WV_), 0, s>>>(SGL_ARGS(a));                                  \
    else sg::rollout_kernel<G_, WV_, false, false><<<grid, dim3(64 * WV_), 0, s>>>(SGL_ARGS(a))
    SGL_DISPATCH(G, WV, CALL);
#undef CALL
}
} // namespace sgl
