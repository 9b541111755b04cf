// k_wide.hip -- scenarios of more than 512 entities: the step as four kernels over any number of workgroups (sgym_wide.hpp).
#define SG_UNIT_WIDE
#include "sgym_launch.hpp"

namespace sgl {
void wide_step(dim3 ge, dim3 gs, hipStream_t s, const sg::Params &p, double timestep, const sg::WideArgs &wa, bool no_peds)
{
    if (no_peds) {
        sg::wide_move_commit_kernel<<<ge, dim3(256), 0, s>>>(p, timestep, wa);
    } else {
        sg::wide_move_kernel<<<ge, dim3(256), 0, s>>>(p, timestep, wa);
        sg::wide_commit_kernel<<<ge, dim3(256), 0, s>>>(p, timestep, wa);
    }
    sg::wide_collide_kernel<<<dim3(ge.x * ge.x, ge.y), dim3(256), 0, s>>>(p, wa); // (entity tile x slot tile)
    sg::wide_finish_kernel<<<gs, dim3(256), 0, s>>>(p, timestep, wa);
}
void wide_running(hipStream_t s, const sg::Params &p, int *host_word) { sg::wide_running_kernel<<<dim3(1), dim3(256), 0, s>>>(p, host_word); }
} // namespace sgl
