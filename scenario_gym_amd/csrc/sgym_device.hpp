// sgym_device.hpp -- gfx950 device code of the batched rollout engine.
//
// One wavefront (64 lanes) owns 64/G scenarios ("tiles" of G lanes, lane = entity slot) and keeps
// the whole per-entity state of its scenarios in registers across the time loop.  Per step a lane
//   1. advances its knot-segment cache (union-grid segment for batch-replay lanes, own-knot
//      segment for agent lanes) and evaluates the linear interpolant         [T1, B2, A1]
//   2. integrates the Vehicle/PID controller if it is a controlled lane       [V1, V2]
//   3. updates velocity / distance                                           [P1, P2]
//   4. builds its OBB corners, runs the all-pairs broad phase inside its tile with cross-lane
//      reads, and the exact separating-axis test on the surviving pairs via LDS [G1, G2]
//   5. updates ego metrics, collision events and terminal flags               [M1, M3, M4, P4]
//   6. stores the step-materialised state (coalesced fp64 SoA rows)
// IDs in brackets are the rows of SURVEY.md section 8(a); each device function cites the
// reference file:line it reproduces.  Compiled with -ffp-contract=off: fp64 results are
// bit-identical to the reference's numpy/scipy arithmetic wherever that is IEEE add/mul/div/sqrt;
// the only fused operations are the explicit fma() chains of np.linalg.norm (see sg_norm3).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/sgym.h"

#ifndef SG_WAVES_PER_SIMD
#define SG_WAVES_PER_SIMD 2 // register budget of the rollout kernel: 512 / SG_WAVES_PER_SIMD VGPRs per lane
#endif
// controlled lanes per wavefront the table variant of the rollout kernel serves: one per scenario of the wavefront, at most 4
#ifndef SG_CTL_WAVES
#define SG_CTL_WAVES 4 // control_kernel: <= 128 VGPRs (one of them beside two wavefronts of a table kernel, 168 each)
#endif
#ifndef SG_WAVES_PER_SIMD_PED
#define SG_WAVES_PER_SIMD_PED 2 // pedestrian variant
#endif
#define SG_TAB_LANES(G, WV) ((WV) > 1 || (G) >= 64 ? 1 : ((G) >= 32 ? 2 : 4))
#ifndef SG_WAVES_PER_SIMD_TAB
#define SG_WAVES_PER_SIMD_TAB 2 // same for the variant without in-kernel controllers
#endif

// The device code lives in one part per subject; each part builds on the ones before it.
#include "sgym_experiments.hpp" // instrumentation of experiment builds: empty in the product build
#include "sgym_core.hpp"
#include "sgym_agents.hpp"
#include "sgym_road.hpp"
#include "sgym_crowd.hpp"
#include "sgym_collide.hpp"
#include "sgym_grid.hpp"
#include "sgym_rss.hpp"
#include "sgym_rollout.hpp"
#include "sgym_control.hpp"
#include "sgym_sensors.hpp"
