// k_ctl.hip -- the controller pre-pass: control_kernel, control_kernel_riders, control_kernel_fast.
#define SG_UNIT_CTL
#include "sgym_launch.hpp"

namespace sgl {
void control(int which, dim3 grid, hipStream_t s, const sg::Params &p, double timestep, int n_steps, int first, int k0,
             const double *actions, double *tab, int row0, int metrics)
{
    if (which == CTL_RIDERS) sg::control_kernel_riders<<<grid, dim3(64), 0, s>>>(p, timestep, n_steps, first, k0, actions, tab, row0, metrics);
    else if (which == CTL_FAST) sg::control_kernel_fast<<<grid, dim3(64), 0, s>>>(p, timestep, n_steps, first, k0, actions, tab, row0, metrics);
    else sg::control_kernel<<<grid, dim3(64), 0, s>>>(p, timestep, n_steps, first, k0, actions, tab, row0, metrics);
}
} // namespace sgl
