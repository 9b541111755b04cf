// sgym_launch.hpp -- host-side launchers of the rollout kernel families.
//
// libsgym_hip.so is linked from one object per kernel family (k_*.hip: `make -j` compiles them side by side, and an
// experiment on one family rebuilds one object); every object includes sgym_device.hpp and instantiates only the entry
// points its launcher names.  sgym_hip.hip (the C ABI + the setup / sensor / fix-up kernels) calls the launchers below.
// Tile shapes: WV == 1 with G in {4, 8, 16, 32, 64} lanes per scenario, or G == 64 with WV in {2, 4} wavefronts per
// scenario (the plain variant also 8).
#pragma once
#include "sgym_device.hpp"
#include "sgym_wide.hpp"
#include "sgym_queue.hpp"

namespace sgl {

// what every rollout_kernel* entry point that runs its controllers in the kernel takes
struct RolloutArgs {
    const sg::Params *p;
    double timestep;
    int n_steps, do_reset, force;
    const double *actions; // [n][R][2] or nullptr
    const double *tab;     // controller table planes or nullptr
};

// k_plain.hip: rollout_kernel<G, WV, false, tab>  (WV == 8: rollout_kernel<64, 8, false, false>)
void rollout_plain(int G, int WV, bool tab, dim3 grid, hipStream_t s, const RolloutArgs &a);
// k_ped.hip: rollout_kernel<max(G, 16), WV, true, false> / rollout_kernel_rss_ped (rss)
void rollout_ped(int G, int WV, bool rss, dim3 grid, hipStream_t s, const RolloutArgs &a);
void rollout_ped_rss(int G, int WV, dim3 grid, hipStream_t s, const RolloutArgs &a); // (k_ped_rss.hip, through rollout_ped)
// k_crowd.hip: rollout_kernel_crowd<WV> / rollout_kernel_crowd_models<WV> / rollout_kernel_crowd_riders<WV>
void rollout_crowd(int WV, bool riders, dim3 grid, hipStream_t s, const RolloutArgs &a, bool models = false);
// k_wide.hip (sgym_wide.hpp): one step (mode 0) or State.reset (mode 1 / 2) of scenarios of more than 512 entities, four kernels
void wide_step(dim3 grid_entities, dim3 grid_scenarios, hipStream_t s, const sg::Params &p, double timestep, const sg::WideArgs &wa,
               bool no_peds /* move + commit as one launch */);
void wide_running(hipStream_t s, const sg::Params &p, int *host_word); // scenarios not done yet -> a word of page-locked host memory
// k_rss.hip: rollout_kernel_rss<G, WV> / rollout_kernel_rss_road<G, WV> (road)
void rollout_rss(int G, int WV, bool road, dim3 grid, hipStream_t s, const RolloutArgs &a);
void rollout_rss_road(int G, int WV, dim3 grid, hipStream_t s, const RolloutArgs &a); // (k_rss_road.hip, through rollout_rss)
// k_rss_tab.hip: rollout_kernel_rss_tab<G> + rss_lines_kernel
void rollout_rss_tab(int G, dim3 grid, hipStream_t s, const sg::Params &p, double timestep, int force, const sg::TabGroups &tg);
void rss_lines(dim3 grid, hipStream_t s, const sg::Params &p, const sg::TabGroups &tg);
// k_road.hip: rollout_kernel_road<G, WV>
void rollout_road(int G, int WV, dim3 grid, hipStream_t s, const RolloutArgs &a);
// k_tab.hip: rollout_kernel_tab<G> / rollout_kernel_tab_planar<G>
void rollout_tab(int G, bool planar, dim3 grid, hipStream_t s, const sg::Params &p, double timestep, int force, const sg::TabGroups &tg);
// k_tabq.hip (sgym_queue.hpp): rollout_kernel_tabq<G> / rollout_kernel_tabq_planar<G> -- the table path as one persistent launch
void rollout_tabq(int G, bool planar, dim3 grid, hipStream_t s, const sg::Params &p, double timestep, int force, const sg::TabQueue &tq);
int tabq_waves_per_cu(int G, bool planar); // resident wavefronts of that kernel per compute unit (occupancy query; 0: unknown)
int rss_tabq_waves_per_cu(int G);          // (k_rss_tab.hip)
// k_rss_tab.hip: rollout_kernel_rss_tabq<G> -- the same launch with the RSS callback in the step loop
void rollout_rss_tabq(int G, dim3 grid, hipStream_t s, const sg::Params &p, double timestep, int force, const sg::TabQueue &tq);
// k_slice.hip: rollout_kernel_slice<G> / rollout_kernel_slice_tab<G> (tab != nullptr)
void rollout_slice(int G, dim3 grid, hipStream_t s, const sg::Params &p, double timestep, const sg::SliceArgs &sa, const double *tab);
// k_ctl.hip: the controller pre-pass.  which: 0 control_kernel, 1 control_kernel_riders, 2 control_kernel_fast
enum { CTL_GENERAL = 0, CTL_RIDERS = 1, CTL_FAST = 2 };
void control(int which, dim3 grid, hipStream_t s, const sg::Params &p, double timestep, int n_steps, int first, int k0,
             const double *actions, double *tab, int row0, int metrics);

} // namespace sgl

// CALL(G, WV) for the tile shape (G, WV) of a handle
#define SGL_DISPATCH(G_, WV_, CALL)                                                                                                  \
    do {                                                                                                                             \
        if ((WV_) == 4) { CALL(64, 4); }                                                                                             \
        else if ((WV_) == 2) { CALL(64, 2); }                                                                                        \
        else switch (G_) {                                                                                                           \
            case 4: CALL(4, 1); break;                                                                                               \
            case 8: CALL(8, 1); break;                                                                                               \
            case 16: CALL(16, 1); break;                                                                                             \
            case 32: CALL(32, 1); break;                                                                                             \
            default: CALL(64, 1); break;                                                                                             \
        }                                                                                                                            \
    } while (0)
// CALL(G) for one-wavefront tiles
#define SGL_DISPATCH_G(G_, CALL)                                                                                                     \
    do {                                                                                                                             \
        switch (G_) {                                                                                                                \
        case 4: CALL(4); break;                                                                                                      \
        case 8: CALL(8); break;                                                                                                      \
        case 16: CALL(16); break;                                                                                                    \
        case 32: CALL(32); break;                                                                                                    \
        default: CALL(64); break;                                                                                                    \
        }                                                                                                                            \
    } while (0)
#define SGL_ARGS(a) *(a).p, (a).timestep, (a).n_steps, (a).do_reset, (a).force, (a).actions, (a).tab
