// k_walk.hip -- the walker variant of the crowd rollout (sgym_walk.hpp): walk_classify_kernel + walk_kernel<1 / 2>.
#define SG_UNIT_WALK
#include "sgym_launch.hpp"
#include "sgym_walk4.hpp"

namespace sgl {
void walk_classify(dim3 grid, hipStream_t s, const sg::Params &p, const sg::WalkArgs &wa, int chunk_len, int enable_mask, int walk1_max)
{
    sg::walk_classify_kernel<<<grid, dim3(256), 0, s>>>(p, wa, chunk_len, enable_mask, walk1_max);
}
void walk_rollout(int WVL, dim3 grid, hipStream_t s, const sg::Params &p, double timestep, int n_steps, int force, const sg::WalkArgs &wa)
{
    if (WVL == 4) sg::walk4_kernel<<<grid, dim3(256), 0, s>>>(p, timestep, n_steps, force, wa); // (<= 64 walkers on four wavefronts: sgym_walk4.hpp)
    else if (WVL == 2) sg::walk_kernel<2><<<grid, dim3(128), 0, s>>>(p, timestep, n_steps, force, wa);
    else sg::walk_kernel<1><<<grid, dim3(64), 0, s>>>(p, timestep, n_steps, force, wa);
}
} // namespace sgl
