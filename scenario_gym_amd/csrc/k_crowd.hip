// k_crowd.hip -- all-pedestrian batches (BASELINE config 5): rollout_kernel_crowd<WV>, with several pedestrian models
// rollout_kernel_crowd_models<WV>, and with riders on a pre-pass table rollout_kernel_crowd_riders<WV>.
#include "sgym_launch.hpp"

namespace sgl {
void rollout_crowd(int WV, bool riders, dim3 grid, hipStream_t s, const RolloutArgs &a, bool models)
{
#define CALL(WV_)                                                                                                                    \
    if (riders) sg::rollout_kernel_crowd_riders<WV_><<<grid, dim3(64 * WV_), 0, s>>>(SGL_ARGS(a));                                   \
    else if (models) sg::rollout_kernel_crowd_models<WV_><<<grid, dim3(64 * WV_), 0, s>>>(SGL_ARGS(a));                              \
    else sg::rollout_kernel_crowd<WV_><<<grid, dim3(64 * WV_), 0, s>>>(SGL_ARGS(a))
    if (WV == 4) { CALL(4); }
    else if (WV == 2) { CALL(2); }
    else { CALL(1); }
#undef CALL
}
} // namespace sgl
