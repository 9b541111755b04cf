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

namespace sg {

// static per-entity rows (same 64-slot block layout as the dynamic state, see sgym.h)
enum {
    ST_BW = 0, ST_BL, ST_BCX, ST_BCY,   // BoundingBox width, length, center_x, center_y
    ST_MIN_T, ST_MAX_T,                 // Trajectory.min_t / max_t
    ST_KNOT_OFF,                        // int64 first row of the entity's own knots
    ST_META,                            // int64: kind | etype << 8 | knot_n << 32
    ST_ROUTE,                           // int64: first route waypoint | n_waypoints << 48 (pedestrian agents)
    ST_CTL,                             // int64: column of this slot in the controller table (PID / vehicle agents), -1 = none
    ST_CTRL,                            // NCTRL_ROWS rows of controller parameters (SG_C_*)
    NCTRL_ROWS = 13,
    ST_COUNT = ST_CTRL + NCTRL_ROWS     // 23 rows
};
constexpr uint32_t ROW = 512; // bytes of one field row of a block (64 lanes x 8 B)

struct ScenStatic { // per scenario, read-only
    double t0, length;
    int64_t grid_off;
    int32_t grid_n, ego;
};

// Road surfaces (sg_set_road_networks): the polygons as edge soup + one uniform cell grid per network.
struct RoadNet { double x0, y0, inv_cell; int32_t nx, ny; int64_t cell_base; };
// a polygon whose boundary touches a cell: its edges there, and a reference point of the cell (one of RN_REF below) that is
// not on the polygon's boundary, with its inside / outside status
struct RoadCand { int32_t poly; uint32_t edge_off; uint16_t n_edges; uint8_t ref_sel, ref_inside; };
struct RoadIndex {
    const RoadNet *nets;            // [n_nets]
    const int32_t *net_of_scen;     // [R], -1 = no road network
    const uint16_t *cells;          // per cell: low byte = layers some polygon covers the WHOLE cell with, high byte = layers
                                    // with a polygon whose boundary touches the cell (candidates below)
    const uint32_t *cell_off;       // CSR over all cells of all networks
    const RoadCand *cand;           // the candidates of the cells
    const int32_t *cand_edges;      // their edge lists (indices into edges)
    const double *edges;            // [n_edges][4] x1, y1, x2, y2; the rings of a polygon are contiguous
    const uint32_t *poly_layers;    // [n_polygons] SG_LAYER_*
    const uint32_t *net_flags;      // [n_nets] bit 0: the walkable surface has area, bit 1: the impenetrable surface has
    const int64_t *imp_off;         // [n_nets + 1] ranges of imp_edges
    const double *imp_edges;        // [n][4] ring edges of the impenetrable polygons (buildings), for the nearest-point search
    int32_t n_nets;
};

struct Params {
    int R, E, EP;
    int persist;
    unsigned term_mask;
    int rec_cap, ev_cap;
    const double *stat;      // [n_blocks][ST_COUNT][64]
    const ScenStatic *sstat; // [R]
    const double *knots;     // [rows][7] own knots of every entity
    const double *grid_t;    // union knot grids, all scenarios
    double *grid_y;          // [grid rows][6][EP] stage-1 resample
    double *dyn;             // [n_blocks][FROWS][64]
    sg_scenario_state *sdyn; // [R]
    sg_event *events;        // [R][ev_cap]
    double *ev_pose;         // [R][ev_cap][3] ego x, y, heading at the event (input of classify_events_kernel)
    double *ev_hpose;        // [R][ev_cap][3] the hazard's x, y, heading at the event when it is a controlled agent (its pose
                             // cannot be re-derived from a trajectory); NaN: not saved
    double *rec_t, *rec_pose;
    const double *routes;    // [rows][2] pedestrian route waypoints
    const double *gon;       // [64][2] cos, sin of 2*pi*i/64 (host libm): Point.buffer(r) vertices
    int WV, FROWS;           // waves per scenario (1, 2, 4); rows per state block = SG_F_COLL + WV
    sg_social_force sf;
    // controller pre-pass (control_kernel): the PID / vehicle agents of the whole batch, 64 to a wavefront
    const int32_t *ctl_ent;  // [n_ctl_pad] padded entity index r*EP + slot of controlled lane q, -1 = padding
    double *ctl_state;       // [CS_COUNT][n_ctl_pad] lane state carried from one chunk of steps to the next
    int n_ctl_pad;           // multiple of 64
    const double *ext_pose;  // [NE][6] poses of the caller-run agents (SG_KIND_AGENT_EXTERNAL), x = NaN: agent returned None
    int32_t *rss_state, *rss_code, *rss_seen; // RSSDistances records (sg_rss_update / rollout_kernel_rss), nullptr before first use
    double *rss_safe;
    double *rssq;      // [wavefronts][rssq_cap][RSSQ_REC] line-test queues of rollout_kernel_rss (rss_lines_kernel)
    int32_t *rssq_n;   // [wavefronts] groups queued by the latest launch
    int32_t rssq_cap;  // groups per wavefront
    const unsigned char *reset_mask; // [R] sg_reset_scenarios: the scenarios a do_reset == 2 launch resets
    const RoadIndex *road;   // device copy of the road index, nullptr = no road networks set
    int ctl_general;         // 1: control_kernel without its straight-line fast path (env SG_CTL_FAST=0; the tests compare the two)
    int ped_serial;          // 1: pedestrian pair loop one pedestrian per lane (env SG_PED_SERIAL; default 0: balanced over the wavefront)
    int tab_steps;           // steps per table chunk (rows per lane = tab_steps + 1: the prefetch of the last step reads one row ahead)
    // random fluctuations of the social force (sg_set_ped_noise): 0 off, 1 stream of standard normal variates per scenario,
    // 2 counter-based generator
    int noise_mode;
    double noise_std_lon, noise_std_lat;
    const double *noise_normals; // [R][noise_len]
    long long noise_len;
    unsigned long long noise_seed;
#ifdef SG_PHASE_TIMERS
    unsigned long long *phase_cycles; // [16] experiment builds: s_memtime cycles per phase of the step, summed over wavefronts
#endif
};

// Experiment builds (-DSG_PHASE_TIMERS, tools/ab_build.sh): where do the cycles of a step go?  PH(i) closes phase i: the
// cycles since the previous mark are added to counter i (wave-uniform scalar work); flushed once at the end of the kernel.
#ifdef SG_PHASE_TIMERS
struct PhaseTimers {
    unsigned long long acc[16], last;
    __device__ __forceinline__ void start() { for (int i = 0; i < 16; ++i) acc[i] = 0; last = __builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ void mark(int i) { const unsigned long long now = __builtin_amdgcn_s_memtime(); acc[i] += now - last; last = now; }
    __device__ __forceinline__ void flush(unsigned long long *out) { if ((threadIdx.x & 63) == 0) for (int i = 0; i < 16; ++i) if (acc[i]) atomicAdd(out + i, acc[i]); }
};
#define PH(i) ptm.mark(i)
#else
struct PhaseTimers {};
#define PH(i) ((void)0)
#endif

// controller table written by control_kernel, read by rollout_kernel<.., TAB = true>.  Two planes of
// [n_ctl_pad][tab_steps + 1][4] doubles (the steps of one lane are contiguous: 32 B per step, so the scalar loads of
// two consecutive steps share a cache line):
//   plane 0: x, y, h after step k, controller speed      plane 1: e_lon_prev, e_lat_prev, e_lon_int, unused
//   plane 2 (lanes that are their scenario's ego): EgoAvgSpeed, EgoMaxSpeed, EgoAvgSpeed.t after step k
// Planes 1 and 2 are read once, at the row of the last step the scenario executed.
enum { CT_X = 0, CT_Y, CT_H, CT_SPEED, CT_W = 4, CT_ELON = 0, CT_ELAT, CT_EINT, CT_MAVG = 0, CT_MMAX, CT_MT, CT_PLANES = 3 };
enum { CS_POSE = 0, CS_PRESENT = 6, CS_CTRL = 7, CS_T = 11, CS_PREV_T = 12, CS_METRIC = 13, CS_COUNT = 16 };

// Lane pointers into one 64-slot block.  Global loads/stores carry an immediate offset (the compiler
// only uses 0..4095 of it), so a lane keeps three 64-bit addresses per block -- rows 0-7, 8-15 and
// 16-23 -- and every field access is `address + immediate`: no per-field address registers.  The
// upper two are made opaque to the optimiser, otherwise it re-derives one full 64-bit address per
// field, hoists them all out of the time loop and spills them.
#define SG_GLOBAL __attribute__((address_space(1)))
struct LanePtr {
    SG_GLOBAL char *a[3]; // global address space: global_load/global_store (vmcnt only), never flat_*
    __device__ __forceinline__ LanePtr(const double *blk, uint32_t voff)
    {
        a[0] = (SG_GLOBAL char *)(reinterpret_cast<char *>(const_cast<double *>(blk)) + voff);
        a[1] = a[0] + 8 * ROW;
        a[2] = a[0] + 16 * ROW;
        asm("" : "+v"(a[1]), "+v"(a[2]));
    }
};
// (rows 24 and up -- the collision-row words 4..7 of tiles of 8 wavefronts -- hang off the third address with a larger offset)
template <typename T = double>
__device__ __forceinline__ T fld(const LanePtr &lp, int f)
{
    const int b = f < 24 ? f >> 3 : 2;
    return *reinterpret_cast<SG_GLOBAL const T *>(lp.a[b] + (f - 8 * b) * (int)ROW);
}
template <typename T>
__device__ __forceinline__ void stf(const LanePtr &lp, int f, T v)
{
    const int b = f < 24 ? f >> 3 : 2;
    *reinterpret_cast<SG_GLOBAL T *>(lp.a[b] + (f - 8 * b) * (int)ROW) = v;
}

// ------------------------------------------------------------------------------------------------
// math
// ------------------------------------------------------------------------------------------------
// np.linalg.norm(v[:3]) (state.py:237, metrics/trajectory.py:15-21) = sqrt(v.dot(v)); OpenBLAS' ddot tail
// loop is an FMA chain, reproduced explicitly.
__device__ __forceinline__ double sg_norm3(double a, double b, double c)
{
    return __builtin_sqrt(__builtin_fma(c, c, __builtin_fma(b, b, a * a)));
}
__device__ __forceinline__ double sg_norm2(double a, double b)
{
    return __builtin_sqrt(__builtin_fma(b, b, a * a));
}

// Fixed fp64 sin/cos shared (by restatement) with the CPU oracle: two-step Cody-Waite reduction by
// pi/2 + minimax kernels on [-pi/4, pi/4]; plain add/mul only, so CPU and GPU agree bit-for-bit.
// Stands in for np.sin/np.cos in entity/base.py:113 and controller.py:126-128, 221-226 (<1 ulp).
__device__ __noinline__ double2 sg_sincos_slow(double x)
{
    return make_double2(sin(x), cos(x));
}

// The 16 fp64 coefficients live in constant memory and are fetched with scalar loads at the point of
// use (the table pointer is made opaque once per time step), so they occupy SGPRs for a few dozen
// instructions instead of 32 VGPRs for the whole kernel.
static __constant__ double SG_TRIG[32] = {
    6.36619772367581382433e-01,  // 0 2/pi
    1.57079632673412561417e+00,  // 1 pi/2 head (33 bits)
    6.07710050630396597660e-11,  // 2 pi/2 next 33 bits
    2.02226624879595063154e-21,  // 3 pi/2 tail
    -1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04,  // 4-6 S1..S3
    2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10,   // 7-9 S4..S6
    4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05,   // 10-12 C1..C3
    -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11,  // 13-15 C4..C6
    // 16-28: tan polynomial T0..T12 (|x| < 0.67434)
    3.33333333333334091986e-01, 1.33333333333201242699e-01, 5.39682539762260521377e-02,
    2.18694882948595424599e-02, 8.86323982359930005737e-03, 3.59207910759131235356e-03,
    1.45620945432529025516e-03, 5.88041240820264096874e-04, 2.46463134818469906812e-04,
    7.81794442939557092300e-05, 7.14072491382608190305e-05, -1.85586374855275456654e-05,
    2.59073051863633712884e-05, 0.0, 0.0, 0.0,
};

typedef const __attribute__((address_space(4))) double *ConstTbl; // constant address space: scalar loads

// the kernel of sg_sincos for |x| < 1e5 (callers that have voted the range for the whole wavefront: no branch)
__device__ __forceinline__ void sg_sincos_core(double x, double &s, double &c, ConstTbl K);

__device__ __forceinline__ void sg_sincos(double x, double &s, double &c, ConstTbl K = (ConstTbl)SG_TRIG)
{
    if (!(__builtin_fabs(x) < 1.0e5)) {
        double2 sc = sg_sincos_slow(x);
        s = sc.x;
        c = sc.y;
        return;
    }
    sg_sincos_core(x, s, c, K);
}

__device__ __forceinline__ void sg_sincos_core(double x, double &s, double &c, ConstTbl K)
{
    double fn = __builtin_rint(x * K[0]);
    int n = (int)fn;
    double t = x - fn * K[1];
    double w = fn * K[2];
    double r = t - w;
    w = fn * K[3] - ((t - r) - w);
    double y0 = r - w;
    double y1 = (r - y0) - w;
    double z = y0 * y0;
    double v = z * y0;
    double rs = K[5] + z * (K[6] + z * (K[7] + z * (K[8] + z * K[9])));
    double ks = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * K[4]);
    double rc = z * (K[10] + z * (K[11] + z * (K[12] + z * (K[13] + z * (K[14] + z * K[15])))));
    double ay = __builtin_fabs(y0);
    uint64_t qb = ((uint64_t)__double_as_longlong(ay) - ((uint64_t)0x00200000 << 32)) & 0xFFFFFFFF00000000ULL;
    double qx = ay > 0.78125 ? 0.28125 : __longlong_as_double((long long)qb);
    qx = ay < 0.3 ? 0.0 : qx; // with qx = 0 the two branches of the kernel coincide
    double hz = 0.5 * z - qx;
    double a = 1.0 - qx;
    double kc = a - (hz - (z * rc - y0 * y1));
    double ss = (n & 1) ? kc : ks;
    double cc = (n & 1) ? ks : kc;
    s = (n & 2) ? -ss : ss;
    c = ((n + 1) & 2) ? -cc : cc;
}

// the polynomial branch of sg_tan (|x| < 0.67434)
__device__ __forceinline__ double sg_tan_poly(double x, ConstTbl K)
{
    ConstTbl T = K + 16;
    double z = x * x;
    double w = z * z;
    double r = T[1] + w * (T[3] + w * (T[5] + w * (T[7] + w * (T[9] + w * T[11]))));
    double v = z * (T[2] + w * (T[4] + w * (T[6] + w * (T[8] + w * (T[10] + w * T[12])))));
    double s = z * x;
    r = z * (s * (r + v));
    r = r + T[0] * s;
    return x + r;
}

// tan(steer) of VehicleController._step (controller.py:128): same split as the oracle's sgo_tan
__device__ __forceinline__ double sg_tan(double x, ConstTbl K)
{
    if (!(__builtin_fabs(x) < 0.67434)) {
        double s, c;
        sg_sincos(x, s, c, K);
        return s / c;
    }
    return sg_tan_poly(x, K);
}

// fp32 sin/cos of an fp64 heading for the collision broad phase and filter (never for stored state):
// the angle is reduced to revolutions in fp64 (|error| < 4e-12 rev for |h| < 1e5), rounded to fp32
// (2^-25 rev) and fed to the hardware v_sin_f32 / v_cos_f32, whose argument is in revolutions.
// Absolute error <= SG_TRIG32_ERR; tests/test_gpu_parity.py measures it through sg_debug_trig32.
#define SG_TRIG32_ERR 4.0e-6f
__device__ __forceinline__ void sg_sincos_f32(double h, float &s, float &c)
{
    if (!(__builtin_fabs(h) < 1.0e5)) { // huge / non-finite headings: the fp64 path's own fallback
        double2 sc = sg_sincos_slow(h);
        s = (float)sc.x;
        c = (float)sc.y;
        return;
    }
    const double rev = h * 1.59154943091895345554e-01; // 1 / (2 pi)
    const float f = (float)(rev - __builtin_rint(rev));
    s = __builtin_amdgcn_sinf(f);
    c = __builtin_amdgcn_cosf(f);
}

__device__ __forceinline__ double sg_pred(double x) // nextafter(x, -inf) for finite x
{
    long long b = __double_as_longlong(x);
    if (x > 0.0) return __longlong_as_double(b - 1);
    if (x < 0.0) return __longlong_as_double(b + 1);
    return -4.9406564584124654e-324;
}

// Wavefront votes on the builtin: HIP's __any / __all go through device-library functions (__ockl_wfany_i32) that are
// linked in after the optimiser has run, and every vote on a predicate that already lives in a scalar mask then costs a
// v_cndmask 0/1 + v_cmp round trip through the vector ALU (40 of them in the step loop of rollout_kernel_tab).
__device__ __forceinline__ bool sg_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }
__device__ __forceinline__ bool sg_all(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0; }

__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src, 64); }

// s_waitcnt vmcnt(0) (expcnt / lgkmcnt untouched).  Placed after every RARE block of global loads whose results
// stay in registers across the time loop: the state stores of the steady state share vmcnt with those loads, and
// without an explicit wait at the load site the compiler has to wait for vmcnt(0) -- i.e. for every store of the
// previous step -- at the first use of such a register inside each step.
__device__ __forceinline__ void sg_loads_done() { __builtin_amdgcn_s_waitcnt(0x0F70); }
// s_waitcnt lgkmcnt(0): same idea for LDS / scalar-memory results at the end of a step, so that the scalar
// table loads issued at the top of the next step are not waited for on the spot
__device__ __forceinline__ void sg_lgkm_done() { __builtin_amdgcn_s_waitcnt(0xC07F); }

// ------------------------------------------------------------------------------------------------
// x / d for many numerators and one denominator.  `a / b` on gfx950 expands to
//   v_div_scale x2, v_rcp_f64, 2 Newton steps on the reciprocal, q = a*r, e = a - b*q,
//   v_div_fmas(e, r, q), v_div_fixup
// which is correctly rounded.  When neither operand needs v_div_scale's rescaling (both well inside
// the normal range) that sequence is exactly: r = refined reciprocal of b (depends on b only),
// q0 = a*r, e = fma(-b, q0, a), q = fma(e, r, q0).  RecipDiv hoists the b-only part; callers
// fall back to `/` when an operand is outside the safe range.
// ------------------------------------------------------------------------------------------------
struct RecipDiv {
    double b, r;
    bool ok;
    __device__ __forceinline__ explicit RecipDiv(double den) : b(den)
    {
        double ab = __builtin_fabs(den);
        ok = ab > 0x1p-500 && ab < 0x1p500;
        double r0 = __builtin_amdgcn_rcp(den);
        double e0 = __builtin_fma(-den, r0, 1.0);
        double r1 = __builtin_fma(r0, e0, r0);
        double e1 = __builtin_fma(-den, r1, 1.0);
        r = __builtin_fma(r1, e1, r1);
    }
    // numerator range in which the unscaled sequence is exactly the IEEE quotient: +0, or a biased
    // exponent in [64, 1983] (|a| in [2^-959, 2^961)); denormals, huge values, inf and nan fall back
    __device__ __forceinline__ bool safe(double a) const
    {
        uint32_t e = ((uint32_t)__double2hiint(a) >> 20) & 0x7ffu;
        // (+0 only: -0 / b is -0, the unscaled sequence gives +0)
        return ok & (((e - 64u) < 1920u) | (__double_as_longlong(a) == 0)); // bitwise: straight-line code, no short-circuit branches
    }
    __device__ __forceinline__ double div(double a) const
    {
        double q0 = a * r;
        double e = __builtin_fma(-b, q0, a);
        return __builtin_fma(e, r, q0);
    }
};

// ------------------------------------------------------------------------------------------------
// knot tables.  A lane interpolates either the scenario's union grid (SG_KIND_REPLAY:
// BatchReplayEntity stage 2, entity/batch.py:122-128) or its own knots (agents:
// Trajectory.position_at_t, trajectory.py:142-205).  Both are scipy interp1d(kind="linear"):
//   idx = clip(searchsorted_left(x, t), 1, n-1); slope = (y_hi-y_lo)/(x_hi-x_lo);
//   y = slope*(t-x_lo) + y_lo, with the first/last row outside [x0, x_{n-1}].
// ------------------------------------------------------------------------------------------------
struct Table {
    const double *x;  // times
    const double *y;  // values
    int n;            // rows
    int xs, ys, cs;   // strides (in doubles): x row stride, y row stride, y channel stride
    __device__ __forceinline__ double X(int i) const { return x[(size_t)i * xs]; }
    __device__ __forceinline__ double Y(int i, int c) const { return y[(size_t)i * ys + (size_t)c * cs]; }
};

struct Segment {
    double x_lo, x_hi;
    double ylo[6], sl[6];
    int cur; // 0 = before first knot, 1..n-1 = bracket [cur-1, cur], n = after last knot
};

__device__ __forceinline__ void seg_load(const Table &T, Segment &S)
{
    const int n = T.n, cur = S.cur;
    if (n <= 0) {
        S.x_lo = 0.0;
        S.x_hi = __builtin_inf();
#pragma unroll
        for (int c = 0; c < 6; ++c) { S.ylo[c] = 0.0; S.sl[c] = 0.0; }
        return;
    }
    if (cur == 0 || cur >= n || n == 1) { // constant piece: first or last row
        int row = cur == 0 ? 0 : n - 1;
        double x0 = T.X(row);
        S.x_lo = x0;
        S.x_hi = (cur == 0 && n > 1) ? sg_pred(x0) : __builtin_inf();
#pragma unroll
        for (int c = 0; c < 6; ++c) { S.ylo[c] = T.Y(row, c); S.sl[c] = 0.0; }
        return;
    }
    double x_lo = T.X(cur - 1), x_hi = T.X(cur);
    S.x_lo = x_lo;
    S.x_hi = x_hi;
    double dx = x_hi - x_lo;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        double a = T.Y(cur - 1, c), b = T.Y(cur, c);
        S.ylo[c] = a;
        S.sl[c] = (b - a) / dx;
    }
}

// cursor for time t from scratch (kernel entry)
__device__ __forceinline__ int seg_locate(const Table &T, double t)
{
    const int n = T.n;
    if (n <= 1) return 0;
    if (t < T.X(0)) return 0;
    if (t > T.X(n - 1)) return n;
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (T.X(mid) < t) lo = mid + 1; else hi = mid;
    }
    return lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
}

// advance the cursor to time t (t never decreases inside a launch)
__device__ __forceinline__ void seg_advance(const Table &T, Segment &S, double t)
{
    if (t > S.x_hi) {
        int cur = S.cur == 0 ? 1 : S.cur;
        while (cur <= T.n - 1 && T.X(cur) < t) ++cur;
        S.cur = cur;
        seg_load(T, S);
    }
}

// Trajectory.position_at_t(t, extrapolate=True) on a lane's own knots (trajectory.py:142-205);
// used for a newcomer's previous pose (state.py:219-222) and at reset.
__device__ __forceinline__ void own_position_extrap(const double *kn, int n, double t, double (&out)[6])
{
    if (n == 1) { // trajectory.py:175-177: knot duplicated at t + 1e-3
        double x_lo = kn[0];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            double slope = kn[1 + c] - kn[1 + c]; // (y - y)/(x_hi - x_lo): +0, or NaN for a non-finite knot
            out[c] = slope * (t - x_lo) + kn[1 + c];
        }
        return;
    }
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (kn[(size_t)mid * 7] < t) lo = mid + 1; else hi = mid;
    }
    int idx = lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
    const double *a = kn + (size_t)(idx - 1) * 7, *b = kn + (size_t)idx * 7;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        double slope = (b[1 + c] - a[1 + c]) / (b[0] - a[0]);
        out[c] = slope * (t - a[0]) + a[1 + c];
    }
}

// ------------------------------------------------------------------------------------------------
// geometry
// ------------------------------------------------------------------------------------------------
// Entity.get_bounding_box_points (entity/base.py:100-138): RR, FR, FL, RL
__device__ __forceinline__ void sg_corners(double x, double y, double s, double c, double W, double L,
                                           double cx, double cy, double *o)
{
    double pxm = cx - 0.5 * L, pxp = cx + 0.5 * L, pyp = cy + 0.5 * W, pym = cy - 0.5 * W;
    double ns = -s;
    o[0] = x + (pxm * c + pyp * ns); o[1] = y + (pxm * s + pyp * c);
    o[2] = x + (pxp * c + pyp * ns); o[3] = y + (pxp * s + pyp * c);
    o[4] = x + (pxp * c + pym * ns); o[5] = y + (pxp * s + pym * c);
    o[6] = x + (pxm * c + pym * ns); o[7] = y + (pxm * s + pym * c);
}

// closed-set intersection of two convex quads (shapely `intersects`, utils.py:52-59): separated iff
// one of the 8 edge lines has every vertex of the other quad strictly on its outer side.
__device__ __forceinline__ bool sg_sat_pass(const double *P, const double *Q)
{
    double o = (P[4] - P[0]) * (P[7] - P[3]) - (P[5] - P[1]) * (P[6] - P[2]);
    bool sep = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int j = (i + 1) & 3;
        double ax = P[2 * i], ay = P[2 * i + 1];
        double ex = P[2 * j] - ax, ey = P[2 * j + 1] - ay;
        bool all_out = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double cr = ex * (Q[2 * k + 1] - ay) - ey * (Q[2 * k] - ax);
            bool out = o > 0 ? (cr < 0) : (cr > 0);
            all_out = all_out && out;
        }
        sep = sep || all_out;
    }
    return sep;
}
__device__ __forceinline__ bool sg_quads_intersect(const double *A, const double *B)
{
    return !(sg_sat_pass(A, B) || sg_sat_pass(B, A));
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// Workgroup-shared tile data.  NS = slots of the tile set a workgroup owns: 64 when one wavefront
// carries 64/G scenarios (WV = 1), 64*WV when WV wavefronts carry one scenario of up to 64*WV entities.
template <int NS, bool PED, bool CROWD = false>
struct TileLds {
    // ---- collision scratch: rewritten by every tile_collisions call and dead once it returns (its last reads sit
    // before its last workgroup barrier).  Contiguous, in this order: the pedestrian pair balancer (ped_pairs_balanced)
    // borrows the block between two collision passes. ----
    float cx[NS], cy[NS];   // box centres (NaN when absent), SoA for packed-fp32 pair math
    float2 sc[NS];          // sin, cos of the heading
    float2 cen[NS];         // box centres again, interleaved, for single-read gathers
    // broad-phase stripe masks: bit set of the slots whose centre lies in x- (y-) stripe k (mod 64)
    unsigned long long xtab[64][NS / 64], ytab[64][NS / 64];
    // fp64 corners of the exact path: a single wavefront exchanges them with cross-lane reads instead;
    // the 8 floats that remain are scratch for the launch-time reductions across wavefronts
    double cor[8][NS > 64 ? NS : 2];
    // ---- end of the collision scratch ----
    float2 half[NS];        // half length, half width (static)
    int last[NS];
    int vote[4][8];         // block_vote: one row per vote site, one word per wavefront (tiles of up to 8 wavefronts)
    // controller parameters of every slot, copied once per launch: the 9 vehicle / PID rows, or -- in pedestrian
    // scenes -- the 4 pedestrian rows SG_C_PED_* (index q - SG_C_PED_SPEED_DESIRED)
    double ctrl[PED ? 4 : 9][NS];
    double boxwl[2][NS];    // bounding box width, length (exact path and controllers only)
    // social force inputs of the CURRENT state (pedestrian/sensor.py:55-64): reference point, velocity
    double px[PED ? NS : 1], py[PED ? NS : 1], vx[PED ? NS : 1], vy[PED ? NS : 1];
    // per NEIGHBOUR terms of the repulsion force, computed once by the neighbour itself (social_force.py:148-155):
    // unit velocity o = v / (|v| + 1e-10) and step = (|v| + 1e-10) * (next_t - t)
    double ox[PED ? NS : 1], oy[PED ? NS : 1], stp[PED ? NS : 1];
    unsigned char isped[PED ? NS : 1]; // entity.type == "Pedestrian" and present
    // all-pedestrian scenes (rollout_kernel_crowd): more per-NEIGHBOUR products hoisted out of the pair (stp * o, stp * stp),
    // the thresholds of the radius rule of every pedestrian (r*r*(1 + 1e-9), r*r*0.9975: sg_in_radius) and, per lane, the
    // non-empty 32-bit words of its neighbour candidate row (crowd_pairs walks them as a queue), word-major: the bank of
    // an access depends on the lane only
    double sx[CROWD ? NS : 1], sy[CROWD ? NS : 1], ss[CROWD ? NS : 1];
    double r2hi[CROWD ? NS : 1], r2lo[CROWD ? NS : 1];
    uint32_t nq[CROWD ? 8 : 1][CROWD ? NS : 1];

    static constexpr int SLOTS = NS;
    static constexpr int SCRATCH_BYTES = NS * 40 + 64 * (NS > 64 ? NS : 2); // cx ... cor
    // pairs one wavefront can hand over to its idle lanes: 4 B (who, whom, flags) + 16 B (result) each
    static constexpr int PAIR_CAP = NS > 64 ? 320 : 128;
    __device__ __forceinline__ char *wave_scratch(int wave) { return reinterpret_cast<char *>(cx) + wave * (PAIR_CAP * 20); }
};
static_assert(TileLds<256, true>::PAIR_CAP * 20 * 4 <= TileLds<256, true>::SCRATCH_BYTES, "pair scratch");
static_assert(TileLds<128, true>::PAIR_CAP * 20 * 2 <= TileLds<128, true>::SCRATCH_BYTES, "pair scratch");
static_assert(TileLds<64, true>::PAIR_CAP * 20 <= TileLds<64, true>::SCRATCH_BYTES, "pair scratch");
typedef TileLds<256, true> TileLdsWide;
static_assert(offsetof(TileLdsWide, half) == TileLdsWide::SCRATCH_BYTES, "collision scratch is contiguous");

// ------------------------------------------------------------------------------------------------
// controllers
// ------------------------------------------------------------------------------------------------
struct CtrlState { double speed, e_lon_prev, e_lat_prev, e_lon_int; };

// VehicleController._step (controller.py:105-140); sin_h/cos_h of the current heading come from
// the caller.  cp(q) = controller parameter SG_C_q of this slot (LDS table, or the static rows in pedestrian scenes).
template <typename CP>
__device__ __forceinline__ void vehicle_step(CtrlState &cs, const CP &cp, double l,
                                             double dt, double accel, double steer, double sin_h,
                                             double cos_h, double *pose, ConstTbl K)
{
    double max_steer = cp(SG_C_MAX_STEER), max_accel = cp(SG_C_MAX_ACCEL);
    double max_speed = cp(SG_C_MAX_SPEED), allow_rev = cp(SG_C_ALLOW_REVERSE);
    accel = __builtin_fmin(__builtin_fmax(accel, -max_accel), max_accel);
    steer = __builtin_fmin(__builtin_fmax(steer, -max_steer), max_steer);
    double dx = cs.speed * cos_h;
    double dy = cs.speed * sin_h;
    double dh = cs.speed * sg_tan(steer, K) / l;
    pose[0] += dx * dt;
    pose[1] += dy * dt;
    pose[3] += dh * dt;
    double speed = cs.speed + accel * dt;
    if (allow_rev == 0.0) speed = __builtin_fmax(0.0, speed);
    if (max_speed == max_speed) speed = __builtin_fmin(max_speed, speed);
    cs.speed = speed;
}

// PIDController._step (controller.py:205-258)
template <typename CP>
__device__ __forceinline__ void pid_step(CtrlState &cs, const CP &cp, double l,
                                         double state_dt, double dt, double tx, double ty,
                                         double sin_h, double cos_h, double *pose, ConstTbl K)
{
    double e0 = tx - pose[0], e1 = ty - pose[1];
    double e_lon = cos_h * e0 + sin_h * e1;
    double e_lat = -sin_h * e0 + cos_h * e1;
    double speed = cs.speed, gain;
    if (speed > 5.0 && speed <= 15) gain = 1.0 - 0.9 * (speed - 5.0) / 10.0;
    else if (speed > 15) gain = 0.1;
    else gain = 1.0;
    const RecipDiv rd(state_dt); // both derivative terms divide by State.dt
    const bool fast = rd.safe(e_lat - cs.e_lat_prev) && rd.safe(e_lon - cs.e_lon_prev);
    double e_lat_D = fast ? rd.div(e_lat - cs.e_lat_prev) : (e_lat - cs.e_lat_prev) / state_dt;
    double kp = cp(SG_C_STEER_KP) * gain, kd = cp(SG_C_STEER_KD) * gain;
    double steer = kp * e_lat + kd * e_lat_D;
    double e_lon_D = fast ? rd.div(e_lon - cs.e_lon_prev) : (e_lon - cs.e_lon_prev) / state_dt;
    double e_lon_I = cs.e_lon_int + e_lon * state_dt;
    double accel = 0.0;
    if (__builtin_fabs(e_lon) > 0.1)
        accel = cp(SG_C_ACCEL_KP) * e_lon + cp(SG_C_ACCEL_KD) * e_lon_D + cp(SG_C_ACCEL_KI) * e_lon_I;
    cs.e_lat_prev = e_lat;
    cs.e_lon_prev = e_lon;
    cs.e_lon_int = e_lon_I;
    vehicle_step(cs, cp, l, dt, accel, steer, sin_h, cos_h, pose, K);
}

// ------------------------------------------------------------------------------------------------
// pedestrians: exp / atan2 shared (by restatement) with the oracle's sgo_exp / sgo_atan2
// ------------------------------------------------------------------------------------------------
// Division policy of the social-force pair terms.  Exact: plain IEEE '/'.  Fast: the same quotients through
// RecipDiv (correctly rounded inside its operand range); an operand outside the range only raises `bad`, and the
// caller recomputes that pair with Exact.  Keeps the common case free of branches.
struct ExactArith {
    bool bad = false;
    __device__ __forceinline__ double div(double a, double d) { return a / d; }
    __device__ __forceinline__ void div2(double a, double b, double d, double &qa, double &qb) { qa = a / d; qb = b / d; }
    // fl(a / m) >= c
    __device__ __forceinline__ bool quotient_ge(double a, double m, double c) { return a / m >= c; }
    __device__ __forceinline__ double sqrt(double x) { return __builtin_sqrt(x); }
};
struct FastArith {
    bool bad = false;
    // The compiler's fp64 sqrt is v_rsq_f64 + two Goldschmidt refinements + two residual corrections, wrapped in a
    // 2^256 rescaling for arguments below 2^-767 and a pass-through for 0 / inf.  For arguments in [2^-700, 2^1000) the
    // rescaling is the identity, so the bare core below returns the same bits with 10 instructions instead of 18.
    __device__ __forceinline__ double sqrt(double x)
    {
        bad |= !((x >= 0x1p-700) & (x < 0x1p1000));
        const double y = __builtin_amdgcn_rsq(x);
        double g = x * y, h = y * 0.5;
        const double r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g);
        h = __builtin_fma(h, r, h);
        double d = __builtin_fma(-g, g, x);
        g = __builtin_fma(d, h, g);
        d = __builtin_fma(-g, g, x);
        return __builtin_fma(d, h, g);
    }
    __device__ __forceinline__ double div(double a, double d)
    {
        const RecipDiv rd(d);
        bad |= !rd.safe(a);
        return rd.div(a);
    }
    __device__ __forceinline__ void div2(double a, double b, double d, double &qa, double &qb)
    {
        const RecipDiv rd(d);
        bad |= !(rd.safe(a) & rd.safe(b));
        qa = rd.div(a);
        qb = rd.div(b);
    }
    // m > 0.  Rounding is monotone: a >= c*m*(1 + 2^-50) implies fl(a/m) >= c, a <= c*m*(1 - 2^-50) implies
    // fl(a/m) < c (8 ulp margins); the sliver in between (and c*m outside the normal range) is `bad`.
    __device__ __forceinline__ bool quotient_ge(double a, double m, double c)
    {
        const double cm = c * m, acm = __builtin_fabs(cm), slack = acm * 0x1p-50;
        const bool yes = a >= cm + slack, no = a <= cm - slack;
        bad |= !((yes | no) & (acm < 0x1p1000) & ((acm > 0x1p-900) | (c == 0.0)));
        return yes;
    }
};

template <typename AR>
__device__ __forceinline__ double sg_exp(double x, AR &A)
{
    const double LN2HI = 6.93147180369123816490e-01, LN2LO = 1.90821492927058770002e-10,
                 INVLN2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                 P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                 P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 709.782712893383973096) return __builtin_inf();
    if (x < -745.13321910194110842) return 0.0;
    double k = __builtin_rint(x * INVLN2);
    double hi = x - k * LN2HI;
    double lo = k * LN2LO;
    double r = hi - lo;
    double t = r * r;
    double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    double y = 1.0 - ((lo - A.div(r * c, 2.0 - c)) - hi);
    return ldexp(y, (int)k);
}
__device__ __forceinline__ double sg_exp(double x)
{
    ExactArith A;
    return sg_exp(x, A);
}

// One neighbour's terms of SocialForce._step (social_force.py:59-62, 140-188, 213-222): the weighted repulsion
// (c1) and attraction (c2) this neighbour adds to the force, in the reference's operation order.
// STRAIGHT: head rotation 0 in every lane (hs == 0, hc == 1): the rotated velocity is the velocity itself, so the view
// direction is the neighbour's own unit velocity (odx, ody), bit for bit.  NOATT: ped_attract_C == 0 with a positive
// sight weight: the attraction is a signed zero and w2 * (+-0) == +-0, whatever w2 is.
template <bool STRAIGHT, bool NOATT, typename AR>
__device__ __forceinline__ void ped_pair(AR &A, const sg_social_force &sf, double k2_scale, double px, double py,
                                         double hs, double hc, double ox, double oy, double ovx, double ovy,
                                         double odx, double ody, double step, double &c1x, double &c1y,
                                         double &c2x, double &c2y)
{
    // view direction = the neighbour's velocity rotated by the head angle (:59-62, X.dot(R.T))
    double ux = odx, uy = ody;
    if (!STRAIGHT) {
        double vx = __builtin_fma(ovx, hc, ovy * (-hs)), vy = __builtin_fma(ovx, hs, ovy * hc);
        double vn = A.sqrt(__builtin_fma(vy, vy, vx * vx)) + 0.0000000001;
        A.div2(vx, vy, vn, ux, uy);
    }
    double rx = px - ox, ry = py - oy; // _force_pedestrian_repulsion, :140-176
    double rn = A.sqrt(__builtin_fma(ry, ry, rx * rx));
    double qx = rx - step * odx, qy = ry - step * ody;
    double qn = A.sqrt(__builtin_fma(qy, qy, qx * qx)) + 0.0000000001;
    double sum = rn + qn;
    double b = (1.0 / 2) * A.sqrt(sum * sum - step * step);
    double k1 = (1.0 / 4) * A.div(1.0, b) * sum;
    double rxn, ryn, qxn, qyn;
    A.div2(rx, ry, rn, rxn, ryn);
    A.div2(qx, qy, qn, qxn, qyn);
    double dbx = k1 * (rxn + qxn), dby = k1 * (ryn + qyn);
    double k2 = k2_scale * sg_exp(A.div(-b, sf.ped_repulse_sigma), A);
    double repx = k2 * dbx, repy = k2 * dby;
    double k3 = 2 * sf.ped_attract_C; // _force_pedestrian_attraction, :178-188
    double attx = k3 * rx, atty = k3 * ry;
    double w1 = 1.0, w2 = 1.0;
    if (sf.sight_weight_use != 0.0) { // _sight_weight, :213-222 (wave-uniform)
        w1 = A.quotient_ge(__builtin_fma(uy, repy, ux * repx), A.sqrt(__builtin_fma(repy, repy, repx * repx)) + 0.0000000001, sf.cos_sight)
                 ? 1.0 : sf.sight_weight;
        c1x = w1 * repx; c1y = w1 * repy;
        if (NOATT) {
            c2x = attx; c2y = atty;
        } else {
            w2 = A.quotient_ge(__builtin_fma(uy, atty, ux * attx), A.sqrt(__builtin_fma(atty, atty, attx * attx)) + 0.0000000001, sf.cos_sight)
                     ? 1.0 : sf.sight_weight;
            c2x = w2 * attx; c2y = w2 * atty;
        }
    } else {
        c1x = repx; c1y = repy;
        c2x = attx; c2y = atty;
    }
}

// log for the Box-Muller transform of the counter-based noise generator: fdlibm's __ieee754_log restated, domain finite
// normal x > 0; the same operation sequence as the oracle's sgo_log.
__device__ __forceinline__ double sg_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    int hx = __double2hiint(x);
    int k = (hx >> 20) - 1023;
    hx &= 0x000fffff;
    const int i0 = (hx + 0x95f64) & 0x100000;
    x = __hiloint2double(hx | (i0 ^ 0x3ff00000), __double2loint(x)); // normalize x or x/2
    k += i0 >> 20;
    const double f = x - 1.0, dk = (double)k;
    if ((0x000fffff & (2 + hx)) < 3) { // |f| < 2**-20
        if (f == 0.0) return k == 0 ? 0.0 : dk * ln2_hi + dk * ln2_lo;
        const double R = f * f * (0.5 - 0.33333333333333333 * f);
        return k == 0 ? f - R : dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    const double s = f / (2.0 + f), z = s * s, w = z * z;
    const int i = (hx - 0x6147a) | (0x6b851 - hx);
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6)), t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    if (i > 0) {
        const double hfsq = 0.5 * f * f;
        return k == 0 ? f - (hfsq - s * (hfsq + R)) : dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    return k == 0 ? f - s * (f - R) : dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

// The two standard normal variates of noise mode 2 for (seed, scenario, entity, step): Philox4x32-10 (Salmon et al., SC'11)
// at counter (entity, step, 0, 0) under key (seed_lo ^ scenario, seed_hi), two 53-bit uniforms in (0, 1), Box-Muller.
// Same operation sequence as the oracle's sgo_noise_pair.
__device__ __forceinline__ void sg_noise_pair(unsigned long long seed, uint32_t scenario, uint32_t entity, uint32_t step,
                                              double &z0, double &z1, ConstTbl K)
{
    uint32_t c0 = entity, c1 = step, c2 = 0, c3 = 0, k0 = (uint32_t)seed ^ scenario, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const double u1 = ((double)((((uint64_t)c0 << 32) | c1) >> 11) + 0.5) * 0x1p-53;
    const double u2 = ((double)((((uint64_t)c2 << 32) | c3) >> 11) + 0.5) * 0x1p-53;
    const double r = __builtin_sqrt(-2.0 * sg_log(u1));
    double sn, cs;
    sg_sincos(6.28318530717958623200e+00 * u2, sn, cs, K);
    z0 = r * cs;
    z1 = r * sn;
}

__device__ __forceinline__ double sg_atan_pos(double ax)
{
    const double A0 = 3.33333333333329318027e-01, A1 = -1.99999999998764832476e-01,
                 A2 = 1.42857142725034663711e-01, A3 = -1.11111104054623557880e-01,
                 A4 = 9.09088713343650656196e-02, A5 = -7.69187620504482999495e-02,
                 A6 = 6.66107313738753120669e-02, A7 = -5.83357013379057348645e-02,
                 A8 = 4.97687799461593236017e-02, A9 = -3.65315727442169155270e-02,
                 A10 = 1.62858201153657823623e-02;
    if (ax >= 7.378697629483821e19) return 1.57079632679489655800e+00 + 6.12323399573676603587e-17;
    int id;
    double x, hi, lo;
    if (ax < 0.4375) { id = -1; x = ax; hi = 0.0; lo = 0.0; }
    else if (ax < 0.6875) { id = 0; x = (2.0 * ax - 1.0) / (2.0 + ax); hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17; }
    else if (ax < 1.1875) { id = 1; x = (ax - 1.0) / (ax + 1.0); hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17; }
    else if (ax < 2.4375) { id = 2; x = (ax - 1.5) / (1.0 + 1.5 * ax); hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17; }
    else { id = 3; x = -1.0 / ax; hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17; }
    double z = x * x, w = z * z;
    double s1 = z * (A0 + w * (A2 + w * (A4 + w * (A6 + w * (A8 + w * A10)))));
    double s2 = w * (A1 + w * (A3 + w * (A5 + w * (A7 + w * A9))));
    if (id < 0) return x - x * (s1 + s2);
    return hi - ((x * (s1 + s2) - lo) - x);
}

__device__ __forceinline__ double sg_atan2(double y, double x)
{
    const double PI = 3.1415926535897931160E+00, PI_LO = 1.2246467991473531772E-16;
    if (x != x || y != y) return x + y;
    if (y == 0.0) return (x < 0.0 || (x == 0.0 && __builtin_signbit(x))) ? __builtin_copysign(PI, y) : y;
    if (x == 0.0) return __builtin_copysign(0.5 * PI, y);
    double z = sg_atan_pos(__builtin_fabs(y / x));
    if (x > 0.0) return y > 0.0 ? z : -z;
    z = PI - (z - PI_LO);
    return y > 0.0 ? z : -z;
}

// State.get_entities_in_radius (state/state.py:356-372): centre strictly inside the 64-gon
// Point(cx, cy).buffer(r); gon = cos/sin table of the polygon's vertex angles.
__device__ __forceinline__ bool sg_in_radius(double cx, double cy, double r, double px, double py, const double *gon)
{
    double dx = px - cx, dy = py - cy, d2 = dx * dx + dy * dy, r2 = r * r;
    if (d2 > r2 * (1.0 + 1e-9)) return false;
    if (d2 < r2 * 0.9975) return true;
    // On the thin ring between the inscribed circle and the vertices only the edges facing the point can
    // cut it off: test the edge of its sector and both neighbours with the oracle's cross product (the
    // other 61 edges hold with a margin of ~r*sin(pi/32)).  Vertices run clockwise: (cx + r*C_i, cy - r*S_i).
    float phi = atan2f((float)(-dy), (float)dx);
    int k0 = (int)__builtin_floorf(phi * 10.185916f); // 64 / (2*pi)
    bool inside = true;
    for (int e = -1; e <= 1; ++e) {
        int i = (k0 + e) & 63, j = (i + 1) & 63;
        double ax = cx + r * gon[2 * i], ay = cy - r * gon[2 * i + 1];
        double bx = cx + r * gon[2 * j], by = cy - r * gon[2 * j + 1];
        double cr = (bx - ax) * (py - ay) - (by - ay) * (px - ax);
        inside = inside && (cr < 0);
    }
    return inside;
}

// LineString(route).project(Point) + the goal update of PedestrianAgent._step (pedestrian/agent.py:59-62)
__device__ __forceinline__ int ped_goal_update(const double *wp, int nwp, double px, double py)
{
    double best = __builtin_inf(), best_s = 0.0, acc = 0.0;
    for (int i = 0; i + 1 < nwp; ++i) {
        double ax = wp[2 * i], ay = wp[2 * i + 1], dx = wp[2 * i + 2] - ax, dy = wp[2 * i + 3] - ay;
        double L2 = dx * dx + dy * dy;
        double u = L2 == 0.0 ? 0.0 : __builtin_fmin(1.0, __builtin_fmax(0.0, ((px - ax) * dx + (py - ay) * dy) / L2));
        double qx = ax + u * dx, qy = ay + u * dy;
        double ex = px - qx, ey = py - qy;
        double dist = __builtin_sqrt(ex * ex + ey * ey);
        double L = __builtin_sqrt(L2);
        if (dist < best) { best = dist; best_s = acc + u * L; }
        acc += L;
    }
    double arc = 0.0;
    int last = 0;
    for (int k = 0; k < nwp; ++k) {
        if (k > 0) {
            double dx = wp[2 * k] - wp[2 * k - 2], dy = wp[2 * k + 1] - wp[2 * k - 1];
            arc += __builtin_sqrt(dx * dx + dy * dy);
        }
        if (arc <= best_s) last = k;
    }
    return last + 1;
}

// ------------------------------------------------------------------------------------------------
// Road surfaces: point strictly inside the union of the polygons of a layer.
// shapely contains(Point) (state.py:401-407, sensor/map.py:198-271) = JTS/GEOS RayCrossingCounter: the ray towards +x
// crosses the polygon's rings an odd number of times; a point ON a ring is not contained.  The orientation sign is
// exact: fp64 determinant with Shewchuk's stage-A error bound, else the six products of the expanded determinant as
// two-term expansions, summed exactly (grow-expansion); the sign of the sum is the sign of its largest component.
// Host and device share these functions (the host uses them to classify the grid cells, sgym_hip.hip).
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline void rn_two_sum(double a, double b, double &s, double &e)
{
    const double x = a + b, bb = x - a;
    s = x;
    e = (a - (x - bb)) + (b - bb);
}

__host__ __device__ __attribute__((noinline)) inline int rn_orient_exact(double ax, double ay, double bx, double by, double px, double py)
{
    // (ax - px)(by - py) - (ay - py)(bx - px) = ax*by - ax*py - px*by - ay*bx + ay*px + py*bx
    const double fa[6] = {ax, -ax, -px, -ay, ay, py}, fb[6] = {by, py, by, bx, px, bx};
    double e[12];
    int n = 0;
    for (int k = 0; k < 6; ++k) {
        const double hi = fa[k] * fb[k], lo = __builtin_fma(fa[k], fb[k], -hi);
        for (int u = 0; u < 2; ++u) {
            double q = u ? hi : lo;
            for (int i = 0; i < n; ++i) rn_two_sum(q, e[i], q, e[i]);
            e[n++] = q;
        }
    }
    for (int i = n - 1; i >= 0; --i)
        if (e[i] != 0.0) return e[i] > 0 ? 1 : -1;
    return 0;
}

__host__ __device__ inline int rn_orient_sign(double ax, double ay, double bx, double by, double px, double py)
{
    const double dl = (ax - px) * (by - py), dr = (ay - py) * (bx - px), det = dl - dr;
    const double bound = 1e-15 * (__builtin_fabs(dl) + __builtin_fabs(dr));
    if (det > bound) return 1;
    if (det < -bound) return -1;
    return rn_orient_exact(ax, ay, bx, by, px, py);
}

// RayCrossingCounter.countSegment: toggles `cross` on a crossing, returns true if the point is ON the edge
__host__ __device__ inline bool rn_ray_edge(double x1, double y1, double x2, double y2, double px, double py, bool &cross)
{
    if (x1 < px && x2 < px) return false;
    if (px == x2 && py == y2) return true;
    if (y1 == py && y2 == py) {
        const double lo = x1 < x2 ? x1 : x2, hi = x1 < x2 ? x2 : x1;
        return px >= lo && px <= hi;
    }
    if ((y1 > py && y2 <= py) || (y2 > py && y1 <= py)) {
        int o = rn_orient_sign(x1, y1, x2, y2, px, py);
        if (o == 0) return true;
        if (y2 < y1) o = -o;
        if (o > 0) cross = !cross;
    }
    return false;
}

// 0 outside, 1 strictly inside, 2 on a ring -- the whole polygon (host: cell classification and reference points)
__host__ __device__ inline int rn_polygon_locate(const double *edges, int64_t e0, int64_t e1, double px, double py)
{
    bool cross = false;
    for (int64_t i = e0; i < e1; ++i) {
        const double *e = edges + i * 4;
        if (rn_ray_edge(e[0], e[1], e[2], e[3], px, py, cross)) return 2;
    }
    return cross ? 1 : 0;
}

// cell of a point; false = outside the grid (the grid covers every polygon with a margin, so: outside every surface)
__host__ __device__ inline bool rn_cell_of(const RoadNet &N, double px, double py, int &ix, int &iy)
{
    const double fx = (px - N.x0) * N.inv_cell, fy = (py - N.y0) * N.inv_cell;
    if (!(fx >= 0.0 && fx < (double)N.nx && fy >= 0.0 && fy < (double)N.ny)) return false;
    ix = (int)fx;
    iy = (int)fy;
    return true;
}

// candidate reference points of a cell (fractions of the cell side; cell = 1 / inv_cell is a power of two, so the
// products are exact and host and device agree bit for bit)
#define RN_NREF 8
__host__ __device__ inline void rn_ref_point(const RoadNet &N, int ix, int iy, int sel, double &x, double &y)
{
    const double FX[RN_NREF] = {0.5, 0.25, 0.75, 0.25, 0.75, 0.375, 0.625, 0.4375};
    const double FY[RN_NREF] = {0.5, 0.25, 0.25, 0.75, 0.75, 0.5625, 0.3125, 0.6875};
    const double c = 1.0 / N.inv_cell;
    x = N.x0 + ((double)ix + FX[sel]) * c;
    y = N.y0 + ((double)iy + FY[sel]) * c;
}

// Inside a cell whose reference point R has a known status: P has the same status unless the segment R -> P crosses the
// polygon's boundary an odd number of times, and only edges that touch the cell can cross a segment inside it.
// Crossing of edge (a, b): a and b on different sides of the line R-P (half-open: "left of" vs "not left of", so a
// boundary passing through a vertex counts once) and R, P on different sides of the line a-b.  Returns 0 outside,
// 1 inside, 2 = P lies on one of the edges.
__host__ __device__ inline int rn_locate_in_cell(const double *edges, const int32_t *list, int n, double rx, double ry,
                                                 bool r_inside, double px, double py)
{
    bool inside = r_inside;
    for (int j = 0; j < n; ++j) {
        const double *e = edges + (int64_t)list[j] * 4;
        const double ax = e[0], ay = e[1], bx = e[2], by = e[3];
        const int o2 = rn_orient_sign(ax, ay, bx, by, px, py);
        if (o2 == 0 && px >= (ax < bx ? ax : bx) && px <= (ax < bx ? bx : ax) && py >= (ay < by ? ay : by) && py <= (ay < by ? by : ay))
            return 2;
        const bool sa = rn_orient_sign(rx, ry, px, py, ax, ay) > 0, sb = rn_orient_sign(rx, ry, px, py, bx, by) > 0;
        if (sa != sb) {
            const int o1 = rn_orient_sign(ax, ay, bx, by, rx, ry);
            if ((o1 > 0) != (o2 > 0)) inside = !inside;
        }
    }
    return inside ? 1 : 0;
}

// the layers of `want` whose union strictly contains the point (one thread)
__device__ inline uint32_t rn_layers_at(const RoadIndex &R, int net, uint32_t want, double px, double py)
{
    if (net < 0) return 0u;
    const RoadNet N = R.nets[net];
    int ix, iy;
    if (!rn_cell_of(N, px, py, ix, iy)) return 0u;
    const int64_t cell = N.cell_base + (int64_t)iy * N.nx + ix;
    const uint32_t m = R.cells[cell];
    uint32_t in = m & 0xffu & want, todo = (m >> 8) & want & ~in;
    if (todo) {
        for (uint32_t k = R.cell_off[cell]; k < R.cell_off[cell + 1] && todo; ++k) {
            const RoadCand cd = R.cand[k];
            const uint32_t L = R.poly_layers[cd.poly] & todo;
            if (!L) continue;
            double rx, ry;
            rn_ref_point(N, ix, iy, cd.ref_sel, rx, ry);
            if (rn_locate_in_cell(R.edges, R.cand_edges + cd.edge_off, cd.n_edges, rx, ry, cd.ref_inside != 0, px, py) == 1) {
                in |= L;
                todo &= ~L;
            }
        }
    }
    return in;
}

// The boundary terms of SocialForce._step (pedestrian/social_force.py:86-104, _force_boundary :190-211) for one
// pedestrian at (px, py) of scenario r.  nearest_points(surface, point) is GEOS DistanceOp: a point inside (or on) an
// areal geometry is its own nearest point, so the walkable term -- evaluated only INSIDE the walkable surface -- is the
// zero vector (+0.0 is still added, as the reference does), and so is the impenetrable term inside a building (-0.0);
// outside, every ring edge of the buildings in order: Distance::pointToSegment, nearest first on ties,
// LineSegment::closestPoint.  Same operation sequence as the oracle.
__device__ inline void ped_boundary_terms(const Params &p, int r, double px, double py, double &fx, double &fy)
{
    if (!p.road) return;
    const RoadIndex RI = *p.road;
    const int net = RI.net_of_scen[r];
    if (net < 0) return;
    const uint32_t flags = RI.net_flags[net];
    if (!flags) return;
    const uint32_t in = rn_layers_at(RI, net, SG_LAYER_WALKABLE | SG_LAYER_IMPENETRABLE, px, py);
    if ((flags & 1u) && (in & SG_LAYER_WALKABLE)) { fx += 0.0; fy += 0.0; }
    if (!(flags & 2u)) return;
    if (in & SG_LAYER_IMPENETRABLE) { fx += -0.0; fy += -0.0; return; }
    double best = __builtin_inf(), cx = px, cy = py;
    for (int64_t i = RI.imp_off[net]; i < RI.imp_off[net + 1]; ++i) {
        const double *e = RI.imp_edges + i * 4;
        const double ax = e[0], ay = e[1], bx = e[2], by = e[3];
        auto dist = [](double x0, double y0, double x1, double y1) {
            const double dx = x0 - x1, dy = y0 - y1;
            return __builtin_sqrt(dx * dx + dy * dy);
        };
        double d;
        if (ax == bx && ay == by) {
            d = dist(px, py, ax, ay);
        } else {
            const double len2 = (bx - ax) * (bx - ax) + (by - ay) * (by - ay);
            const double rr = ((px - ax) * (bx - ax) + (py - ay) * (by - ay)) / len2;
            if (rr <= 0.0) d = dist(px, py, ax, ay);
            else if (rr >= 1.0) d = dist(px, py, bx, by);
            else d = __builtin_fabs(((ay - py) * (bx - ax) - (ax - px) * (by - ay)) / len2) * __builtin_sqrt(len2);
        }
        if (d < best) {
            best = d;
            double f;
            if (px == ax && py == ay) f = 0.0;
            else if (px == bx && py == by) f = 1.0;
            else {
                const double dx = bx - ax, dy = by - ay, len = dx * dx + dy * dy;
                f = len <= 0.0 ? __builtin_nan("") : ((px - ax) * dx + (py - ay) * dy) / len;
            }
            if (f > 0.0 && f < 1.0) { cx = ax + f * (bx - ax); cy = ay + f * (by - ay); }
            else if (dist(ax, ay, px, py) < dist(bx, by, px, py)) { cx = ax; cy = ay; }
            else { cx = bx; cy = by; }
        }
    }
    const double rx = px - cx, ry = py - cy, rn = sg_norm2(rx, ry);
    const double ux = rx / (rn + 0.0000000001), uy = ry / (rn + 0.0000000001);
    const double k = p.sf.imp_boundary_repulse_U / p.sf.imp_boundary_repulse_R, ex = sg_exp(-rn / p.sf.imp_boundary_repulse_R);
    fx += 1.0 * (k * ux * ex);
    fy += 1.0 * (k * uy * ex);
}

// Barrier between the lanes of one tile's workgroup.  A single wavefront (WV == 1) needs no s_barrier and no
// s_waitcnt: the LDS executes the instructions of one wavefront in issue order, so a ds_read issued after another
// lane's ds_write / ds_or already sees it; a wavefront-scope fence keeps the compiler from reordering them.
template <int WV>
__device__ __forceinline__ void tile_sync()
{
    if (WV == 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

template <int WV>
__device__ __forceinline__ bool block_any(bool x)
{
    if (WV == 1) return sg_any(x);
    return __syncthreads_or(x);
}

// OR of a two-bit value over the workgroup with ONE barrier (__syncthreads_or is three barriers and an LDS atomic): every
// wavefront leaves its own OR in its word of the site's row, the barrier publishes the row, everybody reads it.  A row is
// rewritten only by the next use of the same site, and every wavefront has read the row before it reaches any later
// barrier -- callers keep at least one barrier between two uses of a site (tile_collisions opens with one).
template <int WV, typename LDS>
__device__ __forceinline__ int block_vote(LDS &L, int site, bool b0, bool b1 = false)
{
    const int mine = (sg_any(b0) ? 1 : 0) | (sg_any(b1) ? 2 : 0);
    if (WV == 1) return mine;
    if ((threadIdx.x & 63) == 0) L.vote[site][threadIdx.x >> 6] = mine;
    __syncthreads();
    int r = 0;
#pragma unroll
    for (int w = 0; w < WV; ++w) r |= L.vote[site][w];
    return r;
}

// One (pedestrian, neighbour) pair: the PedestrianSensor filter (pedestrians only, inside the radius, sensor.py:55-64)
// and the neighbour's two force terms.  (ipx, ipy, irad, hs, hc) describe the pedestrian the force acts on, j is the
// neighbour's LDS slot.  A candidate that fails the filter still runs through the arithmetic (its lane would idle
// anyway) and is masked by the returned flag: fewer branches in a loop that is bound by instruction issue.
template <typename LDS>
__device__ __forceinline__ bool ped_pair_eval(const Params &p, const LDS &L, bool plain, double k2_scale, double ipx,
                                              double ipy, double irad, double hs, double hc, int j, bool valid,
                                              double &c1x, double &c1y, double &c2x, double &c2y)
{
    const sg_social_force &sf = p.sf;
    const double ox = L.px[j], oy = L.py[j];
    const bool act = valid & (L.isped[j] != 0) & sg_in_radius(ipx, ipy, irad, ox, oy, p.gon);
    const double ovx = L.vx[j], ovy = L.vy[j];
    const double odx = L.ox[j], ody = L.oy[j], step = L.stp[j];
    FastArith FA;
    if (plain) // wave-uniform: default head rotation and no attraction
        ped_pair<true, true>(FA, sf, k2_scale, ipx, ipy, hs, hc, ox, oy, ovx, ovy, odx, ody, step, c1x, c1y, c2x, c2y);
    else
        ped_pair<false, false>(FA, sf, k2_scale, ipx, ipy, hs, hc, ox, oy, ovx, ovy, odx, ody, step, c1x, c1y, c2x, c2y);
    if (sg_any(FA.bad & act)) { // rare: some operand outside RecipDiv's range, or a sight weight on its threshold
        if (FA.bad & act) {
            ExactArith EA;
            ped_pair<false, false>(EA, sf, k2_scale, ipx, ipy, hs, hc, ox, oy, ovx, ovy, odx, ody, step, c1x, c1y, c2x, c2y);
        }
    }
    return act;
}

// SocialForce._step :64-84: the neighbour's terms join the force in the reference's order
__device__ __forceinline__ void ped_accumulate(const sg_social_force &sf, double c1x, double c1y, double c2x, double c2y,
                                               double &fx, double &fy)
{
    if (sf.sight_weight_use != 0.0) {
        fx += c1x; fy += c1y;
        fx += c2x; fy += c2y;
    } else { // without sight weights the reference adds the attraction first (:72-80)
        fx += c2x; fy += c2y;
        fx += c1x; fy += c1y;
    }
}

// Neighbour loop, one pedestrian per lane: neighbours in entity order, one per iteration across all row words (the
// wavefront iterates max-over-lanes of the TOTAL candidate count, not the sum of per-word maxima).
template <int WV, typename LDS>
__device__ __forceinline__ void ped_pairs_serial(const Params &p, const LDS &L, int tile0, const uint64_t (&nbr)[WV],
                                                 bool go, bool plain, double k2_scale, double ipx, double ipy,
                                                 double irad, double hs, double hc, double &fx, double &fy)
{
    uint64_t m[WV];
#pragma unroll
    for (int w = 0; w < WV; ++w) m[w] = go ? nbr[w] : 0;
    for (;;) {
        int j = -1;
#pragma unroll
        for (int w = WV - 1; w >= 0; --w)
            if (m[w]) j = w * 64 + __builtin_ctzll(m[w]);
        if (j < 0) break;
#pragma unroll
        for (int w = 0; w < WV; ++w)
            if ((j >> 6) == w) m[w] &= m[w] - 1;
        double c1x, c1y, c2x, c2y;
        if (ped_pair_eval(p, L, plain, k2_scale, ipx, ipy, irad, hs, hc, j + tile0, true, c1x, c1y, c2x, c2y))
            ped_accumulate(p.sf, c1x, c1y, c2x, c2y, fx, fy);
    }
}

// The same sums with the pairs of one wavefront spread evenly over its 64 lanes.  A crowd gives the lanes of a wavefront
// very different neighbour counts (mean ~24, maximum ~45 in the 1024 x 256 benchmark) and the serial loop runs the
// maximum.  Here every lane works through T = ceil(total / 64) pairs: a lane with n > T neighbours keeps its first
// n - o (entity order) and lists the last o in LDS; lanes with n < T (and lanes that are no stepping pedestrian at
// all) evaluate listed pairs for their owners and leave the two force terms in LDS; each owner then adds the terms it
// handed over, in entity order, after its own.  Every pair goes through the same ped_pair_eval and every sum keeps
// the reference's order, so the result is bit-identical to ped_pairs_serial.  Only the "plain" case (no head
// rotation, no attraction: c2 is a signed zero, kept as a sign bit) -- the reference's defaults.
// Wave-collective: all 64 lanes call it in uniform control flow; LDS traffic stays inside the wavefront's own
// slice of the (then idle) collision scratch, so no workgroup barrier is involved.
template <int WV, typename LDS>
__device__ __forceinline__ void ped_pairs_balanced(const Params &p, LDS &L, int sl, int tile0, const uint64_t (&nbr)[WV],
                                                   bool go, double k2_scale, double ipx, double ipy, double irad,
                                                   double &fx, double &fy)
{
    constexpr int CAP = LDS::PAIR_CAP;
    const int lane = threadIdx.x & 63;
    uint32_t *list = reinterpret_cast<uint32_t *>(L.wave_scratch(WV == 1 ? 0 : (int)(threadIdx.x >> 6)));
    double2 *res = reinterpret_cast<double2 *>(list + CAP);
    uint64_t m[WV];
    int n = 0;
#pragma unroll
    for (int w = 0; w < WV; ++w) {
        m[w] = go ? nbr[w] : 0;
        n += __builtin_popcountll(m[w]);
    }
    int total = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) total += __shfl_xor(total, o, 64);
    const int T = (total + 63) >> 6;
    const int excess = max(n - T, 0), spare = max(T - n, 0);
    int scan = excess | (spare << 16); // both prefix sums at once (each < 2^15)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int u = __shfl_up(scan, o, 64);
        if (lane >= o) scan += u;
    }
    const int listed_all = min(__shfl(scan, 63, 64) & 0xffff, CAP);
    const int e0 = (scan & 0xffff) - excess;              // first list position of this lane's hand-over
    const int out = min(max(CAP - e0, 0), excess);        // pairs handed over (all of the excess unless the list is full)
    int h = min((scan >> 16) - spare, listed_all);        // listed pairs this lane evaluates: [h, h_end)
    const int h_end = min((scan >> 16), listed_all);
    // hand over the LAST `out` neighbours: walk them from the top, write them in entity order
    for (int q = 0; sg_any(q < out); ++q) {
        if (q < out) {
            int j = 0;
#pragma unroll
            for (int w = 0; w < WV; ++w)
                if (m[w]) j = w * 64 + 63 - __builtin_clzll(m[w]);
#pragma unroll
            for (int w = 0; w < WV; ++w)
                if ((j >> 6) == w) m[w] &= ~(1ull << (j & 63));
            list[e0 + out - 1 - q] = (uint32_t)(j + tile0) | ((uint32_t)lane << 10) /* slot in bits 0..9 (tiles of up to 512 slots), owner lane above */;
        }
    }
    tile_sync<1>();
    const int wave_sl = sl - lane; // LDS slot of lane 0
    for (;;) {
        int j = -1;
#pragma unroll
        for (int w = WV - 1; w >= 0; --w)
            if (m[w]) j = w * 64 + __builtin_ctzll(m[w]);
        const bool own = j >= 0, help = !own & (h < h_end);
        if (!sg_any(own | help)) break;
#pragma unroll
        for (int w = 0; w < WV; ++w)
            if ((j >> 6) == w) m[w] &= m[w] - 1; // j = -1 matches no word
        const int hi = min(h, CAP - 1);
        const uint32_t ent = list[hi];
        const int isl = wave_sl + (int)((ent >> 10) & 63);
        const int jj = own ? j + tile0 : (int)(ent & (LDS::SLOTS - 1));
        const double qx = own ? ipx : L.px[isl], qy = own ? ipy : L.py[isl];
        const double qr = own ? irad : L.ctrl[SG_C_PED_RADIUS - SG_C_PED_SPEED_DESIRED][isl];
        double c1x, c1y, c2x, c2y;
        const bool act = ped_pair_eval(p, L, true, k2_scale, qx, qy, qr, 0.0, 1.0, jj, own | help, c1x, c1y, c2x, c2y);
        if (own & act) ped_accumulate(p.sf, c1x, c1y, c2x, c2y, fx, fy);
        if (help) {
            res[hi] = make_double2(c1x, c1y);
            list[hi] = ent | (act ? 0u : 1u << 16) | (__builtin_signbit(c2x) ? 1u << 17 : 0u) |
                       (__builtin_signbit(c2y) ? 1u << 18 : 0u);
            ++h;
        }
    }
    tile_sync<1>();
    for (int q = 0; sg_any(q < out); ++q) {
        if (q < out) {
            const uint32_t ent = list[e0 + q];
            const double2 c1 = res[e0 + q];
            if (!(ent & (1u << 16)))
                ped_accumulate(p.sf, c1.x, c1.y, (ent & (1u << 17)) ? -0.0 : 0.0, (ent & (1u << 18)) ? -0.0 : 0.0, fx, fy);
        }
    }
    tile_sync<1>(); // the collision pass that follows rewrites the scratch
}

// ------------------------------------------------------------------------------------------------
// All-pedestrian scenes (rollout_kernel_crowd, BASELINE config 5).
//
// crowd_pair is ped_pair<true, true> (default head rotation, no attraction, sight weights on: the reference's defaults)
// with FastArith's operation sequence -- bit for bit -- but (i) the products that depend on the neighbour alone
// (stp * o, stp * stp) are read from LDS, computed once by the neighbour itself, and (ii) FastArith's operand range checks
// are replaced by GUARDS that are established once per step for the whole tile (crowd_sane, voted in tile_collisions) and
// once per launch for the parameters (crowd_params_ok), plus four exponent compares per pair.  Why that suffices, for a pair
// that is ACTIVE (inside the radius rule, so |r| <= radius * (1 + 1e-9) < 2^21); inactive pairs are masked, garbage is fine:
//   guards: every coordinate and every product stp * o of a present pedestrian is 0 or has magnitude in [2^-800, 2^400)
//           (coordinates) / [2^-800, 2^20) (products); radius < 2^20; sigma, |cos_sight| in [2^-100, 2^100] (cos_sight may be
//           0); V / sigma <= 2^100.  So rx, ry, qx, qy are 0 or multiples of 2^-852 of magnitude < 2^22: safe numerators of
//           RecipDiv (zero, or |a| in [2^-959, 2^961)).
//   checks: the arguments of the first three square roots are >= 2^-100 (else `bad`): then rn >= 2^-50, qn >= 1e-10,
//           b >= 2^-51 are safe denominators, 1 / b <= 2^51, k1 <= 2^72, |rep| <= 2^173, every sqrt argument is inside
//           [2^-700, 2^1000) where the bare rsq + Goldschmidt core equals the compiler's sqrt (FastArith::sqrt); the argument
//           of the fourth (|rep|^2) is checked against 2^-700.  exp: x = -b / sigma is in [-2^122, -2^-151]; its internal
//           quotient r*c / (2 - c) has 2 - c in (1.6, 2.4) and r*c = 0 or |r*c| >= 2^-302 (k = 0: r = x; k != 0: r is a
//           multiple of 2^-85) -- safe; x < -745.2 returns 0 before the quotient matters.  The sight-weight comparison keeps
//           FastArith's sliver test (`bad` when the quotient is within 8 ulps of cos_sight).
// A `bad` pair is recomputed with plain IEEE divisions (ped_pair<.., ExactArith>) under one wave-uniform branch.
// ------------------------------------------------------------------------------------------------
struct CrowdConsts {
    double k2_scale, sig_b, sig_r; // V / sigma; RecipDiv(sigma)
    double cos_sight, sight_weight, k3;
};

__device__ __forceinline__ double sg_sqrt_core(double x) // FastArith::sqrt without the range check
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}

__device__ __forceinline__ double crowd_exp(double x) // sg_exp for x < 0 finite (see the guards above)
{
    const double LN2HI = 6.93147180369123816490e-01, LN2LO = 1.90821492927058770002e-10,
                 INVLN2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                 P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                 P5 = 4.13813679705723846039e-08;
    const double k = __builtin_rint(x * INVLN2);
    const double hi = x - k * LN2HI;
    const double lo = k * LN2LO;
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    const RecipDiv rd(2.0 - c);
    const double y = 1.0 - ((lo - rd.div(r * c)) - hi);
    const double e = ldexp(y, (int)k);
    return x < -745.13321910194110842 ? 0.0 : e;
}

__device__ __forceinline__ bool crowd_params_ok(const sg_social_force &sf)
{
    const double sg_ = sf.ped_repulse_sigma, ac = __builtin_fabs(sf.cos_sight), k2s = sf.ped_repulse_V / sf.ped_repulse_sigma;
    return sg_ >= 0x1p-100 && sg_ <= 0x1p100 && (ac == 0.0 || (ac >= 0x1p-100 && ac <= 0x1p100)) &&
           __builtin_fabs(k2s) <= 0x1p100 && sf.ped_attract_C == 0.0 && sf.sight_weight > 0.0 && sf.sight_weight_use != 0.0;
}

// the per-entity guard of crowd_pair: c = coordinate / product of a present pedestrian
__device__ __forceinline__ bool crowd_sane(double v, double hi_bound)
{
    const double a = __builtin_fabs(v);
    return (v == 0.0) | ((a >= 0x1p-800) & (a < hi_bound));
}

// (rx, ry) = owner - neighbour.  d2: the squared distance of the radius rule (sg_in_radius: dx*dx + dy*dy, and
// (-a)*(-a) == a*a), sharing the product rx*rx with the first norm.
__device__ __forceinline__ void crowd_pair(const CrowdConsts &C, double rx, double ry, double odx, double ody, double sx,
                                           double sy, double ss, double &c1x, double &c1y, double &c2x, double &c2y,
                                           double &d2, bool &bad)
{
    const double rxx = rx * rx;
    d2 = rxx + ry * ry;
    const double a_rn = __builtin_fma(ry, ry, rxx);
    const double rn = sg_sqrt_core(a_rn);
    const double qx = rx - sx, qy = ry - sy;
    const double a_qn = __builtin_fma(qy, qy, qx * qx);
    const double qn = sg_sqrt_core(a_qn) + 0.0000000001;
    const double sum = rn + qn;
    const double a_b = sum * sum - ss;
    const double b = (1.0 / 2) * sg_sqrt_core(a_b);
    const RecipDiv rb(b);
    const double k1 = (1.0 / 4) * rb.div(1.0) * sum;
    const RecipDiv rrn(rn), rqn(qn);
    const double rxn = rrn.div(rx), ryn = rrn.div(ry), qxn = rqn.div(qx), qyn = rqn.div(qy);
    const double dbx = k1 * (rxn + qxn), dby = k1 * (ryn + qyn);
    RecipDiv rsig(1.0);
    rsig.b = C.sig_b;
    rsig.r = C.sig_r;
    const double k2 = C.k2_scale * crowd_exp(rsig.div(-b));
    const double repx = k2 * dbx, repy = k2 * dby;
    c2x = C.k3 * rx; // the attraction with C == 0: a signed zero
    c2y = C.k3 * ry;
    const double a_rep = __builtin_fma(repy, repy, repx * repx);
    const double m = sg_sqrt_core(a_rep) + 0.0000000001;
    const double a = __builtin_fma(ody, repy, odx * repx);
    const double cm = C.cos_sight * m, slack = __builtin_fabs(cm) * 0x1p-50;
    const bool yes = a >= cm + slack, no = a <= cm - slack;
    const double w1 = yes ? 1.0 : C.sight_weight;
    c1x = w1 * repx;
    c1y = w1 * repy;
    const int h123 = min(min(__double2hiint(a_rn), __double2hiint(a_qn)), __double2hiint(a_b));
    bad = !((h123 >= 0x39B00000) & (__double2hiint(a_rep) >= 0x14300000) & (yes | no)); // 2^-100, 2^-700
}

// crowd_pair for N pairs at once, statement by statement ACROSS the pairs: the instruction stream alternates between N
// independent dependency chains.  A pair is one chain of ~130 dependent fp64 operations (an fp64 result can feed the next
// instruction only ~14 cycles after its issue, 4 cycles apart is the issue rate): written pair after pair the chains stay
// apart in the stream and a wavefront that is alone on its SIMD (sgym_walk.hpp) runs at the latency, not at the issue rate.
// Same operations in the same order per pair: the same bits as crowd_pair.
#define SG_EACH(u) _Pragma("unroll") for (int u = 0; u < N; ++u)
template <int N>
__device__ __forceinline__ void sg_sqrt_core_n(const double (&x)[N], double (&out)[N])
{
    double y[N], g[N], h[N], r[N], d[N];
    SG_EACH(u) y[u] = __builtin_amdgcn_rsq(x[u]);
    SG_EACH(u) { g[u] = x[u] * y[u]; h[u] = y[u] * 0.5; }
    SG_EACH(u) r[u] = __builtin_fma(-h[u], g[u], 0.5);
    SG_EACH(u) { g[u] = __builtin_fma(g[u], r[u], g[u]); h[u] = __builtin_fma(h[u], r[u], h[u]); }
    SG_EACH(u) d[u] = __builtin_fma(-g[u], g[u], x[u]);
    SG_EACH(u) g[u] = __builtin_fma(d[u], h[u], g[u]);
    SG_EACH(u) d[u] = __builtin_fma(-g[u], g[u], x[u]);
    SG_EACH(u) out[u] = __builtin_fma(d[u], h[u], g[u]);
}
// the refined reciprocal of RecipDiv (its b-only part), N at once
template <int N>
__device__ __forceinline__ void sg_recip_n(const double (&den)[N], double (&r)[N])
{
    double r0[N], e0[N], r1[N], e1[N];
    SG_EACH(u) r0[u] = __builtin_amdgcn_rcp(den[u]);
    SG_EACH(u) e0[u] = __builtin_fma(-den[u], r0[u], 1.0);
    SG_EACH(u) r1[u] = __builtin_fma(r0[u], e0[u], r0[u]);
    SG_EACH(u) e1[u] = __builtin_fma(-den[u], r1[u], 1.0);
    SG_EACH(u) r[u] = __builtin_fma(r1[u], e1[u], r1[u]);
}
// RecipDiv::div with the reciprocal r of b: q0 = a r, e = fma(-b, q0, a), q = fma(e, r, q0)
template <int N>
__device__ __forceinline__ void sg_rdiv_n(const double (&a)[N], const double (&b)[N], const double (&r)[N], double (&q)[N])
{
    double q0[N], e[N];
    SG_EACH(u) q0[u] = a[u] * r[u];
    SG_EACH(u) e[u] = __builtin_fma(-b[u], q0[u], a[u]);
    SG_EACH(u) q[u] = __builtin_fma(e[u], r[u], q0[u]);
}
template <int N>
__device__ __forceinline__ void crowd_pair_n(const CrowdConsts &C, const double (&rx)[N], const double (&ry)[N], const double (&odx)[N],
                                             const double (&ody)[N], const double (&sx)[N], const double (&sy)[N], const double (&ss)[N],
                                             double (&c1x)[N], double (&c1y)[N], double (&c2x)[N], double (&c2y)[N], double (&d2)[N],
                                             bool (&bad)[N])
{
    double rxx[N], a_rn[N], rn[N], qx[N], qy[N], a_qn[N], qn[N], sum[N], a_b[N], b[N], rb[N], k1[N], rrn[N], rqn[N];
    double rxn[N], ryn[N], qxn[N], qyn[N], dbx[N], dby[N], one[N], inv_b[N], xarg[N], ex[N], k2[N], repx[N], repy[N], a_rep[N], m[N], a[N];
    SG_EACH(u) rxx[u] = rx[u] * rx[u];
    SG_EACH(u) { d2[u] = rxx[u] + ry[u] * ry[u]; a_rn[u] = __builtin_fma(ry[u], ry[u], rxx[u]); }
    sg_sqrt_core_n<N>(a_rn, rn);
    SG_EACH(u) { qx[u] = rx[u] - sx[u]; qy[u] = ry[u] - sy[u]; }
    SG_EACH(u) a_qn[u] = __builtin_fma(qy[u], qy[u], qx[u] * qx[u]);
    sg_sqrt_core_n<N>(a_qn, qn);
    SG_EACH(u) qn[u] = qn[u] + 0.0000000001;
    SG_EACH(u) sum[u] = rn[u] + qn[u];
    SG_EACH(u) a_b[u] = sum[u] * sum[u] - ss[u];
    sg_sqrt_core_n<N>(a_b, b);
    SG_EACH(u) b[u] = (1.0 / 2) * b[u];
    sg_recip_n<N>(b, rb);
    SG_EACH(u) one[u] = 1.0;
    sg_rdiv_n<N>(one, b, rb, inv_b);
    SG_EACH(u) k1[u] = (1.0 / 4) * inv_b[u] * sum[u];
    sg_recip_n<N>(rn, rrn);
    sg_recip_n<N>(qn, rqn);
    sg_rdiv_n<N>(rx, rn, rrn, rxn);
    sg_rdiv_n<N>(ry, rn, rrn, ryn);
    sg_rdiv_n<N>(qx, qn, rqn, qxn);
    sg_rdiv_n<N>(qy, qn, rqn, qyn);
    SG_EACH(u) { dbx[u] = k1[u] * (rxn[u] + qxn[u]); dby[u] = k1[u] * (ryn[u] + qyn[u]); }
    // rsig.div(-b): the shared reciprocal of sigma
    {
        double q0[N], e[N];
        SG_EACH(u) q0[u] = -b[u] * C.sig_r;
        SG_EACH(u) e[u] = __builtin_fma(-C.sig_b, q0[u], -b[u]);
        SG_EACH(u) xarg[u] = __builtin_fma(e[u], C.sig_r, q0[u]);
    }
    // crowd_exp, N at once
    {
        const double LN2HI = 6.93147180369123816490e-01, LN2LO = 1.90821492927058770002e-10, INVLN2 = 1.44269504088896338700e+00;
        const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                     P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
        double k[N], hi[N], lo[N], r[N], t[N], c[N], den[N], rd[N], rc[N], q[N], y[N];
        SG_EACH(u) k[u] = __builtin_rint(xarg[u] * INVLN2);
        SG_EACH(u) { hi[u] = xarg[u] - k[u] * LN2HI; lo[u] = k[u] * LN2LO; }
        SG_EACH(u) r[u] = hi[u] - lo[u];
        SG_EACH(u) t[u] = r[u] * r[u];
        SG_EACH(u) c[u] = P4 + t[u] * P5;
        SG_EACH(u) c[u] = P3 + t[u] * c[u];
        SG_EACH(u) c[u] = P2 + t[u] * c[u];
        SG_EACH(u) c[u] = P1 + t[u] * c[u];
        SG_EACH(u) c[u] = r[u] - t[u] * c[u];
        SG_EACH(u) den[u] = 2.0 - c[u];
        sg_recip_n<N>(den, rd);
        SG_EACH(u) rc[u] = r[u] * c[u];
        sg_rdiv_n<N>(rc, den, rd, q);
        SG_EACH(u) y[u] = 1.0 - ((lo[u] - q[u]) - hi[u]);
        SG_EACH(u) { const double e_ = ldexp(y[u], (int)k[u]); ex[u] = xarg[u] < -745.13321910194110842 ? 0.0 : e_; }
    }
    SG_EACH(u) k2[u] = C.k2_scale * ex[u];
    SG_EACH(u) { repx[u] = k2[u] * dbx[u]; repy[u] = k2[u] * dby[u]; }
    SG_EACH(u) { c2x[u] = C.k3 * rx[u]; c2y[u] = C.k3 * ry[u]; }
    SG_EACH(u) a_rep[u] = __builtin_fma(repy[u], repy[u], repx[u] * repx[u]);
    sg_sqrt_core_n<N>(a_rep, m);
    SG_EACH(u) m[u] = m[u] + 0.0000000001;
    SG_EACH(u) a[u] = __builtin_fma(ody[u], repy[u], odx[u] * repx[u]);
    SG_EACH(u) {
        const double cm = C.cos_sight * m[u], slack = __builtin_fabs(cm) * 0x1p-50;
        const bool yes = a[u] >= cm + slack, no = a[u] <= cm - slack;
        const double w1 = yes ? 1.0 : C.sight_weight;
        c1x[u] = w1 * repx[u];
        c1y[u] = w1 * repy[u];
        const int h123 = min(min(__double2hiint(a_rn[u]), __double2hiint(a_qn[u])), __double2hiint(a_b[u]));
        bad[u] = !((h123 >= 0x39B00000) & (__double2hiint(a_rep[u]) >= 0x14300000) & (yes | no));
    }
}

#ifndef SG_CROWD_ILP
#define SG_CROWD_ILP 2 // (pedestrian, neighbour) pairs a lane evaluates side by side: independent fp64 dependency chains
#endif

// The neighbour sums of one wavefront of an all-pedestrian scene: ped_pairs_balanced's scheme (every lane works through
// ceil(total / 64) pairs; a lane with more neighbours hands its LAST ones over through LDS, the owner adds the returned
// terms after its own, in entity order: bit-identical to the serial loop) with
//   - the candidate row walked as a queue of its non-empty 32-bit words in LDS (one ffbl + one conditional refill per
//     neighbour instead of a scan over the row's 2 * WV words),
//   - SG_CROWD_ILP pairs per loop round (the pair is one chain of dependent fp64 operations; at two wavefronts per SIMD one
//     chain per wavefront leaves a third of the issue slots empty),
//   - crowd_pair for the arithmetic.
// Wave-collective; LDS traffic stays inside the wavefront's own slice of the (idle) collision scratch + its own nq columns.
template <int WV, typename LDS>
__device__ __forceinline__ void crowd_pairs(const Params &p, LDS &L, const CrowdConsts &C, int sl, const uint64_t (&nbr)[WV],
                                            bool go, double k2_scale, double ipx, double ipy, double &fx, double &fy)
{
    constexpr int CAP = LDS::PAIR_CAP, ND = 2 * WV;
    const int lane = threadIdx.x & 63;
    uint32_t *list = reinterpret_cast<uint32_t *>(L.wave_scratch(WV == 1 ? 0 : (int)(threadIdx.x >> 6)));
    double2 *res = reinterpret_cast<double2 *>(list + CAP);
    // ---- the queue: non-empty words of the row, in order; idxs = their word numbers, 3 bits each ----
    int n = 0, nw = 0;
    uint32_t idxs = 0;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const uint32_t d = go ? (uint32_t)(nbr[i >> 1] >> ((i & 1) * 32)) : 0u;
        if (d) {
            L.nq[nw & 7][sl] = d;
            idxs |= (uint32_t)i << (3 * nw);
            ++nw;
        }
        n += __builtin_popcount(d);
    }
    int total = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) total += __shfl_xor(total, o, 64);
    if (total == 0) return; // wave-uniform
    const int T = (total + 63) >> 6;
    const int excess = max(n - T, 0), spare = max(T - n, 0);
    int scan = excess | (spare << 16); // both prefix sums at once (each < 2^15)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int u = __shfl_up(scan, o, 64);
        if (lane >= o) scan += u;
    }
    const int listed_all = min(__shfl(scan, 63, 64) & 0xffff, CAP);
    const int e0 = (scan & 0xffff) - excess;              // first list position of this lane's hand-over
    const int out = min(max(CAP - e0, 0), excess);        // pairs handed over (all of the excess unless the list is full)
    int h = min((scan >> 16) - spare, listed_all);        // listed pairs this lane evaluates: [h, h_end)
    const int h_end = min((scan >> 16), listed_all);
    const int keep = n - out;
    tile_sync<1>(); // own nq column written above
    // ---- hand over the LAST `out` neighbours: walk the queue from its top, write them in entity order ----
    if (sg_any(out > 0)) {
        int qe = nw - 1;
        uint32_t curh = L.nq[max(qe, 0)][sl];
        for (int q = 0; sg_any(q < out); ++q) {
            if (q < out) {
                const int bit = 31 - __builtin_clz(curh);
                const int j = (int)((idxs >> (3 * qe)) & 7u) * 32 + bit;
                list[e0 + out - 1 - q] = (uint32_t)j | ((uint32_t)lane << 8);
                curh &= ~(1u << bit);
                if (curh == 0) {
                    qe = max(qe - 1, 0);
                    curh = L.nq[qe][sl];
                }
            }
        }
    }
    tile_sync<1>();
    const int wave_sl = sl - lane; // LDS slot of lane 0
    // ---- own pairs first (in order), then listed pairs for other lanes ----
    int k = 0, qi = 0;
    uint32_t cur = L.nq[0][sl];
    const double own_r2hi = L.r2hi[sl], own_r2lo = L.r2lo[sl];
    while (sg_any((k < keep) | (h < h_end))) {
        bool own[SG_CROWD_ILP], help[SG_CROWD_ILP], act[SG_CROWD_ILP], bad[SG_CROWD_ILP], ring[SG_CROWD_ILP];
        int jj[SG_CROWD_ILP], osl[SG_CROWD_ILP], hi_[SG_CROWD_ILP];
        uint32_t ent[SG_CROWD_ILP];
        double c1x[SG_CROWD_ILP], c1y[SG_CROWD_ILP], c2x[SG_CROWD_ILP], c2y[SG_CROWD_ILP];
#pragma unroll
        for (int u = 0; u < SG_CROWD_ILP; ++u) {
            own[u] = k < keep;
            help[u] = !own[u] & (h < h_end);
            // next own neighbour: lowest bit of the current word; an emptied word is replaced by the next of the queue
            const int bit = __builtin_ctz(cur | 0x80000000u);
            const int jo = (int)((idxs >> (3 * qi)) & 7u) * 32 + bit;
            const uint32_t nxt = L.nq[min(qi + 1, 7)][sl];
            const uint32_t rest = cur & (cur - 1);
            const bool adv = own[u] & (rest == 0);
            cur = own[u] ? (adv ? nxt : rest) : cur;
            qi += adv;
            k += own[u];
            hi_[u] = min(h, CAP - 1);
            ent[u] = list[hi_[u]];
            h += help[u];
            jj[u] = own[u] ? jo : (int)(ent[u] & (LDS::SLOTS - 1));
            osl[u] = own[u] ? sl : wave_sl + (int)((ent[u] >> 8) & 63);
        }
#pragma unroll
        for (int u = 0; u < SG_CROWD_ILP; ++u) {
            const int j = jj[u], o = osl[u];
            const double rx = L.px[o] - L.px[j], ry = L.py[o] - L.py[j];
            double d2;
            crowd_pair(C, rx, ry, L.ox[j], L.oy[j], L.sx[j], L.sy[j], L.ss[j], c1x[u], c1y[u], c2x[u], c2y[u], d2, bad[u]);
            const bool valid = own[u] | help[u];
            const bool outside = d2 > L.r2hi[o], inside = d2 < L.r2lo[o];
            ring[u] = valid & !(outside | inside);
            act[u] = valid & inside;
        }
        bool any_ring = false, any_bad = false;
#pragma unroll
        for (int u = 0; u < SG_CROWD_ILP; ++u) any_ring |= ring[u];
        if (sg_any(any_ring)) { // rare: between the inscribed circle and the vertices of the 64-gon Point.buffer(r)
#pragma unroll
            for (int u = 0; u < SG_CROWD_ILP; ++u)
                if (ring[u])
                    act[u] = sg_in_radius(L.px[osl[u]], L.py[osl[u]], L.ctrl[SG_C_PED_RADIUS - SG_C_PED_SPEED_DESIRED][osl[u]],
                                          L.px[jj[u]], L.py[jj[u]], p.gon);
        }
#pragma unroll
        for (int u = 0; u < SG_CROWD_ILP; ++u) any_bad |= bad[u] & act[u];
        if (sg_any(any_bad)) { // rare: an operand outside crowd_pair's range, or a sight weight on its threshold
#pragma unroll
            for (int u = 0; u < SG_CROWD_ILP; ++u)
                if (bad[u] & act[u]) {
                    ExactArith EA;
                    const int j = jj[u], o = osl[u];
                    ped_pair<false, false>(EA, p.sf, k2_scale, L.px[o], L.py[o], 0.0, 1.0, L.px[j], L.py[j], L.vx[j], L.vy[j], L.ox[j],
                                           L.oy[j], L.stp[j], c1x[u], c1y[u], c2x[u], c2y[u]);
                }
        }
#pragma unroll
        for (int u = 0; u < SG_CROWD_ILP; ++u) {
            if (own[u] & act[u]) { // SocialForce._step :64-84 with sight weights: repulsion, then attraction
                fx += c1x[u]; fy += c1y[u];
                fx += c2x[u]; fy += c2y[u];
            }
            if (help[u]) {
                res[hi_[u]] = make_double2(c1x[u], c1y[u]);
                list[hi_[u]] = ent[u] | (act[u] ? 0u : 1u << 16) | (__builtin_signbit(c2x[u]) ? 1u << 17 : 0u) |
                               (__builtin_signbit(c2y[u]) ? 1u << 18 : 0u);
            }
        }
    }
    (void)own_r2hi; (void)own_r2lo; (void)ipx; (void)ipy;
    tile_sync<1>();
    for (int q = 0; sg_any(q < out); ++q) {
        if (q < out) {
            const uint32_t e = list[e0 + q];
            const double2 c1 = res[e0 + q];
            if (!(e & (1u << 16))) {
                fx += c1.x; fy += c1.y;
                fx += (e & (1u << 17)) ? -0.0 : 0.0; fy += (e & (1u << 18)) ? -0.0 : 0.0;
            }
        }
    }
    tile_sync<1>(); // the collision pass that follows rewrites the scratch
}

// PedestrianAgent.step, part 1: SocialForce._step (pedestrian/social_force.py:44-222, boundary terms off) over the
// neighbour candidates `nbr` of the tile.  All inputs are the CURRENT state (LDS px/py/vx/vy).  Wave-collective (every
// lane calls it; `stepping` = this lane is a present pedestrian agent of a running scenario); go = goal not reached yet.
template <int WV, bool CROWD = false, typename LDS>
__device__ __forceinline__ void ped_force(const Params &p, LDS &L, int r, int sl, int tile0, const uint64_t (&nbr)[WV],
                                          bool stepping, const double *pose, double velx, double vely, const double *wp,
                                          int nwp, int &goal_idx, bool &go, double &fx, double &fy, double &vdes,
                                          ConstTbl K, bool crowd_fast = false, const CrowdConsts &CC = CrowdConsts{},
                                          PhaseTimers *ptp = nullptr)
{
#ifdef SG_PHASE_TIMERS
    PhaseTimers ptm_dummy;
    PhaseTimers &ptm = ptp ? *ptp : ptm_dummy;
#endif
    const sg_social_force &sf = p.sf;
    go = false;
    fx = fy = 0.0;
    vdes = 0.0;
    double hs = 0.0, hc = 1.0, radius = 0.0;
    if (stepping) {
        if (goal_idx <= nwp - 1) goal_idx = ped_goal_update(wp, nwp, pose[0], pose[1]);
        if (goal_idx <= nwp - 1) {
            go = true;
            double gx = wp[2 * goal_idx] - pose[0], gy = wp[2 * goal_idx + 1] - pose[1]; // _force_to_goal, :119-138
            double gn = sg_norm2(gx, gy);
            if (gn == 0) gn += 0.000000001;
            vdes = L.ctrl[SG_C_PED_SPEED_DESIRED - SG_C_PED_SPEED_DESIRED][sl];
            const double inv_tau = 1 / sf.relaxation_time;
            fx = inv_tau * (vdes * (gx / gn) - velx);
            fy = inv_tau * (vdes * (gy / gn) - vely);
            if (!CROWD) sg_sincos(L.ctrl[SG_C_PED_HEAD_ROT - SG_C_PED_SPEED_DESIRED][sl], hs, hc, K);
            radius = L.ctrl[SG_C_PED_RADIUS - SG_C_PED_SPEED_DESIRED][sl];
        }
    }
    const double k2_scale = sf.ped_repulse_V / sf.ped_repulse_sigma;
    if (CROWD) {
        if (crowd_fast) { // wave-uniform: the guards of crowd_pair hold
            PH(0);
#ifndef SG_ABL_NO_PAIRS
            crowd_pairs<WV>(p, L, CC, sl, nbr, go, k2_scale, pose[0], pose[1], fx, fy);
#endif
            PH(6);
            return; // (no road network in a crowd launch: no boundary terms)
        }
        if (go) sg_sincos(L.ctrl[SG_C_PED_HEAD_ROT - SG_C_PED_SPEED_DESIRED][sl], hs, hc, K);
    }
    // the shortcuts of ped_pair need the sight-weight branch (c2 = w2 * att) and hold for the whole wavefront
    const bool plain = sg_all(hs == 0.0 && hc == 1.0) && sf.ped_attract_C == 0.0 && sf.sight_weight > 0.0 &&
                       sf.sight_weight_use != 0.0;
#ifdef SG_ABL_NO_PAIRS
    return;
#endif
    if (!CROWD && plain && !p.ped_serial)
        ped_pairs_balanced<WV>(p, L, sl, tile0, nbr, go, k2_scale, pose[0], pose[1], radius, fx, fy);
    else // (CROWD: the guards of crowd_pair do not hold, or SG_PED_SERIAL: the plain serial loop)
        ped_pairs_serial<WV>(p, L, tile0, nbr, go, plain, k2_scale, pose[0], pose[1], radius, hs, hc, fx, fy);
    if (!CROWD && go) ped_boundary_terms(p, r, pose[0], pose[1], fx, fy); // after the neighbours, social_force.py:83-104
}

// PedestrianAgent.step, part 2 (one lane): speed and heading from the force (:110-114, or zero at the goal,
// agent.py:65-68) + PedestrianController._step (pedestrian/controller.py:25-46).
// speed_rand / heading_rand: the random fluctuations np.random.normal(bias, std) of :106-108 (== the bias when std is 0).
__device__ __forceinline__ void ped_move(const Params &p, bool go, double fx, double fy, double vdes, double maxs,
                                         const double *pose, double state_dt, double &cspeed, double &fxo, double &fyo,
                                         double *np_, ConstTbl K, double speed_rand, double heading_rand)
{
    const sg_social_force &sf = p.sf;
    double speed = 0.0, heading = 0.0;
    if (go) {
        speed = __builtin_fmin(sg_norm2(fx, fy) + speed_rand, vdes * sf.max_speed_factor);
        heading = sg_atan2(fy, fx) + heading_rand;
        fxo = fx;
        fyo = fy;
    } else {
        fxo = fyo = 0.0;
    }
    cspeed = __builtin_fmin(__builtin_fmax(speed, -maxs), maxs);
    double hs2, hc2;
    sg_sincos(heading, hs2, hc2, K);
    const double sd = cspeed * state_dt;
#pragma unroll
    for (int c = 0; c < 6; ++c) np_[c] = pose[c];
    np_[0] += sd * hc2;
    np_[1] += sd * hs2;
    np_[3] = heading;
}

// ------------------------------------------------------------------------------------------------
// State.collisions() for one tile (state.py:306-310 -> state/utils.py:10-49 -> utils.py:28-62).
// Fills this lane's adjacency row (bit j of word j/64 = tile slot j) and, for PED, the lane's
// neighbour candidate row for the next step's social force.
//
//   broad phase  fp32 bounding circles about the box centres, all pairs inside the tile: every
//                lane walks the tile's centres through wave-uniform LDS broadcasts, packed fp32.
//   filter       fp32 rectangle-rectangle separating-axis test (4 axes) on the candidate pairs
//                with a conservative error margin: certain-overlap / certain-separation decide.
//   exact        pairs inside the margin (touching, or bit-identical boxes) take the fp64
//                8-edge test on the corners -- the same operation sequence as the CPU oracle.
// The fp32 stages are strictly conservative, so the result equals the fp64 test on every pair.
// With WV > 1 the tile spans WV wavefronts of one workgroup; only the decisions that gate LDS
// writes are workgroup-uniform (block_any), the candidate loops run per wavefront.
// ------------------------------------------------------------------------------------------------
// REFINE (pedestrian variants that can hold entities of very different sizes -- a car among pedestrians): the broad phase
// reaches own radius + the LARGEST radius of the tile, which for a pedestrian next to a car's tile-mate means every
// pedestrian within ~3 m; `hetero` (static per tile, voted at launch) then runs one cheap circle test with the PAIR's radii
// over the candidates before the filter.  Conservative like the broad phase itself, so it cannot change any output.
template <int G, int WV, bool PED, bool CROWD = false, bool REFINE = false, typename LDS>
__device__ __forceinline__ void tile_collisions(bool present, const double *pose, double velx, double vely,
                                                double dtn /* next_t - t of the coming step (PED) */,
                                                double bcx, double bcy, float rad_thr, float trig_eps,
                                                float nbr_thr, float cell_inv, bool is_ped_type, int sl, int tile0, LDS &L,
                                                uint64_t (&rows_out)[WV], uint64_t (&mult_rows)[WV],
                                                uint64_t (&nbr_out)[WV], bool &dense /* in: this lane's wish from the previous call,
                                                out: its wish for the next one; see all_pairs */, bool *crowd_ok = nullptr,
                                                PhaseTimers *ptp = nullptr, bool hetero = false, float rmax_t = 0.0f)
{
#ifdef SG_PHASE_TIMERS
    PhaseTimers ptm_dummy;
    PhaseTimers &ptm = ptp ? *ptp : ptm_dummy;
#endif
    constexpr int TS = G * WV; // tile slots
    const int slot = sl - tile0;
    const double x = pose[0], y = pose[1];
    // box centre in fp32 from the hardware sin/cos; the bounding circle radius and every error margin
    // (fp32 rounding, SG_TRIG32_ERR x centre offset) live in rad_thr (static per lane)
    float fs, fc;
    sg_sincos_f32(pose[3], fs, fc);
    const float bcxf = (float)bcx, bcyf = (float)bcy;
    const float nanf_ = __builtin_nanf("");
    const float fx = present ? (float)x + (bcxf * fc - bcyf * fs) : nanf_;
    const float fy = present ? (float)y + (bcxf * fs + bcyf * fc) : nanf_;
    // fp32 conversion error of the centre grows with |coordinate|: 2^-19 * (|x| + |y|) covers both lanes
    const float mag = __builtin_fabsf(fx) + __builtin_fabsf(fy);
    const float reach = rad_thr + 1.9073486e-6f * mag;
    const float thr = reach * reach;
    const float nreach = nbr_thr + 1.9073486e-6f * mag;
    const float nthr = nreach * nreach;
    // stripe coordinates: cells of side 1/cell_inv >= every reach in the tile, so two slots within reach
    // of each other sit in the same or in adjacent x-stripes AND y-stripes
    const float ax = fx * cell_inv, ay = fy * cell_inv;
    const int ix = present ? (int)__builtin_floorf(ax) : 0, iy = present ? (int)__builtin_floorf(ay) : 0;
    const bool far_out = present && !(__builtin_fabsf(ax) < 4000.0f && __builtin_fabsf(ay) < 4000.0f);
    PH(8); tile_sync<WV>(); PH(11);
    L.cx[sl] = fx;
    L.cy[sl] = fy;
    L.cen[sl] = make_float2(fx, fy);
    L.sc[sl] = make_float2(fs, fc);
    reinterpret_cast<unsigned long long *>(L.xtab)[sl] = 0ull;
    reinterpret_cast<unsigned long long *>(L.ytab)[sl] = 0ull;
    bool insane = false; // CROWD: this lane breaks a guard of crowd_pair
    if (PED) {
        // (CROWD: an absent slot can reach a candidate row through the all-pairs walk, whose masks do not know the presence
        // of other wavefronts' slots; crowd_pairs has no isped test, a NaN position fails its radius rule)
        // (... and a rider that is not a pedestrian -- a car -- is nobody's social-force neighbour, pedestrian/sensor.py:56-63)
        L.px[sl] = (!CROWD || (present && is_ped_type)) ? x : __builtin_nan("");
        L.py[sl] = y; L.vx[sl] = velx; L.vy[sl] = vely;
        L.isped[sl] = present && is_ped_type;
        const double vmag = sg_norm2(velx, vely) + 0.0000000001; // social_force.py:148-155, once per neighbour
        const double uox = velx / vmag, uoy = vely / vmag, stp = vmag * dtn;
        L.ox[sl] = uox;
        L.oy[sl] = uoy;
        L.stp[sl] = stp;
        if (CROWD) { // the neighbour's products of ped_pair, once per neighbour: step * odx, step * ody, step * step
            const double sx = stp * uox, sy = stp * uoy;
            L.sx[sl] = sx;
            L.sy[sl] = sy;
            L.ss[sl] = stp * stp;
            insane = present & !(crowd_sane(x, 0x1p400) & crowd_sane(y, 0x1p400) & crowd_sane(sx, 0x1p20) & crowd_sane(sy, 0x1p20) &
                                 (stp < 0x1p20));
        }
    }
    uint64_t cand[WV];
    bool any_cand = false;
#pragma unroll
    for (int w = 0; w < WV; ++w) { rows_out[w] = 0; mult_rows[w] = 0; nbr_out[w] = 0; cand[w] = 0; }
    // `dense` (workgroup-uniform, pedestrian scenes): a crowd packed tighter than the stripe cells makes almost the
    // whole tile a cell neighbour, and the all-pairs walk below (fixed cost, packed fp32, 4 slots per LDS read) is then
    // cheaper than one circle test per candidate.  Either way the result is a conservative candidate set that the same
    // exact tests refine, so the switch cannot change any output.
    // (CROWD: the same vote also carries the guards of crowd_pair: a scene beyond 4000 cells is no crowd to be fast on;
    // PED: and the broad-phase strategy, which some lane asked for at the end of the previous call)
    PH(2);
    bool odd;
    if (WV == 1 && !PED) {
        odd = sg_any(far_out);
    } else {
        const int voted = block_vote<WV>(L, 0, far_out | insane, PED && dense);
        odd = voted & 1;
        dense = (voted & 2) != 0;
    }
    PH(12);
    if (CROWD) *crowd_ok = !odd;
#if defined(SG_DENSE_NEVER)
    const bool all_pairs = odd;
#elif defined(SG_DENSE_ALWAYS)
    const bool all_pairs = true;
#else
    const bool all_pairs = odd || (PED && dense);
#endif
    if (!all_pairs) { // block_any / the barrier below also publish the LDS writes above
        // ---- stripe masks: O(tile) instead of O(tile^2) ----
        if (WV == 1) tile_sync<WV>();
        const int wsl = (WV == 1) ? 0 : (slot >> 6);              // word of this slot inside the tile's row
        const uint64_t mybit = 1ull << ((WV == 1) ? (sl & 63) : (slot & 63));
        if (present) {
            atomicOr(&L.xtab[ix & 63][wsl], mybit);
            atomicOr(&L.ytab[iy & 63][wsl], mybit);
        }
        PH(2); tile_sync<WV>(); PH(13);
#pragma unroll
        for (int w = 0; w < WV; ++w) {
            uint64_t mx = L.xtab[(ix - 1) & 63][w] | L.xtab[ix & 63][w] | L.xtab[(ix + 1) & 63][w];
            uint64_t my = L.ytab[(iy - 1) & 63][w] | L.ytab[iy & 63][w] | L.ytab[(iy + 1) & 63][w];
            uint64_t m = mx & my;
            if (WV == 1) { // several tiles share the wave: keep this tile's slots, tile-local bit positions
                m >>= tile0;
                if (G < 64) m &= (1ull << (G & 63)) - 1;
            }
            if ((slot >> 6) == w) m &= ~(1ull << (slot & 63)); // not with itself
            cand[w] = present ? m : 0;
        }
        // ---- bounding circles of the cell neighbours: per wavefront, LDS reads only ----
        uint64_t close[WV];
        int iters = 0; // wave-uniform
#pragma unroll
        for (int w = 0; w < WV; ++w) {
            close[w] = 0;
            while (sg_any(cand[w] != 0)) {
                ++iters;
                if (cand[w]) {
                    const int jl = __builtin_ctzll(cand[w]);
                    cand[w] &= cand[w] - 1;
                    const float2 o = L.cen[tile0 + w * 64 + jl];
                    const float dx = o.x - fx, dy = o.y - fy;
                    const float d2 = __builtin_fmaf(dy, dy, dx * dx);
                    if (d2 <= thr) close[w] |= 1ull << jl;
                    if (PED && d2 <= nthr) nbr_out[w] |= 1ull << jl;
                }
            }
            cand[w] = close[w];
            any_cand = any_cand || cand[w] != 0;
        }
        if (PED) dense = iters > (2 * TS) / 5; // ~ where 25 instructions per candidate overtake the walk (voted by the next call)
        PH(9);
    } else {
    // ---- fallback for coordinates beyond 4000 cells: all pairs of the tile ----
    // lane i tests itself against slots j..j+3 per iteration (wave-uniform LDS broadcast reads, one
    // ds_read_b128 per coordinate, two iterations prefetched), everything in packed fp32 (2 columns per
    // v_pk_* op): d2 = dx*dx + dy*dy, then thr - d2 whose SIGN bit says "outside"; the sign bits are
    // shifted into the lane's row with one v_alignbit_b32 per column (columns walked high -> low).
    const v2f fx2 = {fx, fx}, fy2 = {fy, fy}, thr2 = {thr, thr};
    const v2f nthr2 = {nthr, nthr};
    uint32_t out_w[2 * WV], nout_w[2 * WV]; // bit j = 1: slot j is OUTSIDE this lane's reach
#pragma unroll
    for (int w = 0; w < 2 * WV; ++w) { out_w[w] = 0u; nout_w[w] = 0u; }
    if (WV == 1) tile_sync<WV>();
    v4f xs = *reinterpret_cast<const v4f *>(&L.cx[tile0 + TS - 4]);
    v4f ys = *reinterpret_cast<const v4f *>(&L.cy[tile0 + TS - 4]);
    v4f xs1 = xs, ys1 = ys;
    if (TS >= 8) {
        xs1 = *reinterpret_cast<const v4f *>(&L.cx[tile0 + TS - 8]);
        ys1 = *reinterpret_cast<const v4f *>(&L.cy[tile0 + TS - 8]);
    }
    if (PED) {
    // one 32-bit word of the row at a time, both loops unrolled: every index into out_w / nout_w is a constant (a dynamic
    // index would put the two arrays into scratch memory, with a load and a store per group of four slots)
    constexpr int NW32 = (TS + 31) / 32, PER = TS >= 32 ? 8 : TS / 4;
#ifdef SG_ABL_WALK_TWICE // timing experiment: the cost of one walk = the difference to the normal build
    for (int rep_ = 0; rep_ < 2; ++rep_) {
    asm volatile("" : "+v"(xs), "+v"(ys), "+v"(xs1), "+v"(ys1));
#endif
#pragma unroll
    for (int w2 = NW32 - 1; w2 >= 0; --w2) {
        uint32_t w = 0u, v = 0u;
#pragma unroll
        for (int q = PER - 1; q >= 0; --q) {
            const int jb = w2 * 32 + q * 4;
            v4f xs2 = xs1, ys2 = ys1; // two groups of four slots stay in flight
            if (jb >= 8) {
                xs2 = *reinterpret_cast<const v4f *>(&L.cx[tile0 + jb - 8]);
                ys2 = *reinterpret_cast<const v4f *>(&L.cy[tile0 + jb - 8]);
            }
            v2f dxa = v2f{xs.x, xs.y} - fx2, dya = v2f{ys.x, ys.y} - fy2;
            v2f dxb = v2f{xs.z, xs.w} - fx2, dyb = v2f{ys.z, ys.w} - fy2;
            v2f d2a = __builtin_elementwise_fma(dya, dya, dxa * dxa);
            v2f d2b = __builtin_elementwise_fma(dyb, dyb, dxb * dxb);
            v2f ma = thr2 - d2a, mb = thr2 - d2b;
            w = __builtin_amdgcn_alignbit(w, __float_as_uint(mb.y), 31); // w = (w << 1) | sign
            w = __builtin_amdgcn_alignbit(w, __float_as_uint(mb.x), 31);
            w = __builtin_amdgcn_alignbit(w, __float_as_uint(ma.y), 31);
            w = __builtin_amdgcn_alignbit(w, __float_as_uint(ma.x), 31);
            if (PED) { // second reach: PedestrianSensor.distance_threshold
                v2f na = nthr2 - d2a, nb = nthr2 - d2b;
                v = __builtin_amdgcn_alignbit(v, __float_as_uint(nb.y), 31);
                v = __builtin_amdgcn_alignbit(v, __float_as_uint(nb.x), 31);
                v = __builtin_amdgcn_alignbit(v, __float_as_uint(na.y), 31);
                v = __builtin_amdgcn_alignbit(v, __float_as_uint(na.x), 31);
            }
            xs = xs1; ys = ys1;
            xs1 = xs2; ys1 = ys2;
        }
        out_w[w2] = w;
        if (PED) nout_w[w2] = v;
    }
#ifdef SG_ABL_WALK_TWICE
    if (rep_ == 0) {
        xs = *reinterpret_cast<const v4f *>(&L.cx[tile0 + TS - 4]);
        ys = *reinterpret_cast<const v4f *>(&L.cy[tile0 + TS - 4]);
        xs1 = *reinterpret_cast<const v4f *>(&L.cx[tile0 + TS - 8]);
        ys1 = *reinterpret_cast<const v4f *>(&L.cy[tile0 + TS - 8]);
    }
    }
#endif
    } else {
    // (vehicle scenes only come here with coordinates beyond 4000 cells; this form of the loop keeps the table kernels
    // inside their register budget)
#pragma unroll 4
    for (int jb = TS - 4; jb >= 0; jb -= 4) {
        v4f xs2 = xs1, ys2 = ys1; // two groups of four slots stay in flight
        if (jb >= 8) {
            xs2 = *reinterpret_cast<const v4f *>(&L.cx[tile0 + jb - 8]);
            ys2 = *reinterpret_cast<const v4f *>(&L.cy[tile0 + jb - 8]);
        }
        v2f dxa = v2f{xs.x, xs.y} - fx2, dya = v2f{ys.x, ys.y} - fy2;
        v2f dxb = v2f{xs.z, xs.w} - fx2, dyb = v2f{ys.z, ys.w} - fy2;
        v2f d2a = __builtin_elementwise_fma(dya, dya, dxa * dxa);
        v2f d2b = __builtin_elementwise_fma(dyb, dyb, dxb * dxb);
        v2f ma = thr2 - d2a, mb = thr2 - d2b;
        uint32_t w = out_w[jb >> 5];
        w = __builtin_amdgcn_alignbit(w, __float_as_uint(mb.y), 31); // w = (w << 1) | sign
        w = __builtin_amdgcn_alignbit(w, __float_as_uint(mb.x), 31);
        w = __builtin_amdgcn_alignbit(w, __float_as_uint(ma.y), 31);
        w = __builtin_amdgcn_alignbit(w, __float_as_uint(ma.x), 31);
        out_w[jb >> 5] = w;
        xs = xs1; ys = ys1;
        xs1 = xs2; ys1 = ys2;
    }
    }
    // absent slots hold NaN centres (sign bit unspecified): mask them with the tile's presence bits
#pragma unroll
    for (int w = 0; w < WV; ++w) {
        uint64_t pres_w;
        if (WV == 1) {
            pres_w = __ballot(present) >> tile0;
            if (G < 64) pres_w &= (1ull << (G & 63)) - 1;
        } else {
            pres_w = ~0ull; // cross-wave presence: filtered by the NaN-safe compare in the narrow phase
        }
        uint64_t inside = ~(((uint64_t)out_w[2 * w + 1] << 32) | out_w[2 * w]) & pres_w;
        if (WV == 1 && G < 64) inside &= (1ull << (G & 63)) - 1;
        if ((slot >> 6) == w) inside &= ~(1ull << (slot & 63)); // not with itself
        cand[w] = present ? inside : 0;
        if (PED) {
            uint64_t nin = ~(((uint64_t)nout_w[2 * w + 1] << 32) | nout_w[2 * w]) & pres_w;
            if (WV == 1 && G < 64) nin &= (1ull << (G & 63)) - 1;
            if ((slot >> 6) == w) nin &= ~(1ull << (slot & 63));
            nbr_out[w] = present ? nin : 0;
        }
        any_cand = any_cand || cand[w] != 0;
    }
    if (PED) { // back to the stripe masks once nobody has more than TS/12 neighbour candidates (hysteresis)
        int cnt = 0;
#pragma unroll
        for (int w = 0; w < WV; ++w) cnt += __builtin_popcountll(nbr_out[w]);
        dense = cnt > TS / 12; // (a wish: voted by the next call)
        PH(10);
    }
    }
#ifdef SG_ABL_NO_NARROW
#pragma unroll
    for (int w = 0; w < WV; ++w) rows_out[w] = cand[w];
    return;
#endif
    if (REFINE && hetero) { // (uniform over the wavefront / workgroup, fixed for the launch)
        any_cand = false;
#pragma unroll
        for (int w = 0; w < WV; ++w) {
            uint64_t c = cand[w], keep = 0;
            while (sg_any(c != 0)) {
                if (c) {
                    const int jl = __builtin_ctzll(c);
                    c &= c - 1;
                    const int j = tile0 + w * 64 + jl;
                    const float2 o = L.cen[j], oh = L.half[j];
                    const float rj = __builtin_sqrtf(__builtin_fmaf(oh.x, oh.x, oh.y * oh.y)) * 1.00001f; // >= the slot's radius
                    const float dx = o.x - fx, dy = o.y - fy;
                    const float pr = (reach - rmax_t) + rj; // own radius + every margin of `reach` + the other radius
                    if (__builtin_fmaf(dy, dy, dx * dx) <= pr * pr) keep |= 1ull << jl; // (an absent slot: NaN, dropped -- as the filter would)
                }
            }
            cand[w] = keep;
            any_cand = any_cand || keep != 0;
        }
    }
    PH(2);
    // ---- filter: per wavefront, LDS reads only ----
    const float2 myh = L.half[sl];
    const float hl = myh.x, hw = myh.y;
    uint64_t fuzzy[WV];
#pragma unroll
    for (int w = 0; w < WV; ++w) fuzzy[w] = 0;
    bool any_fuzzy = false;
    if (sg_any(any_cand)) {
#pragma unroll
        for (int w = 0; w < WV; ++w) {
            while (sg_any(cand[w] != 0)) {
                if (cand[w]) {
                    const int jl = __builtin_ctzll(cand[w]);
                    cand[w] &= cand[w] - 1;
                    const int j = tile0 + w * 64 + jl;
                    float2 oc = make_float2(L.cx[j], L.cy[j]), os = L.sc[j], oh = L.half[j];
                    float dx = oc.x - fx, dy = oc.y - fy;
                    float cd = __builtin_fabsf(fc * os.y + fs * os.x);  // |cos(delta heading)|
                    float sd = __builtin_fabsf(fs * os.y - fc * os.x);  // |sin(delta heading)|
                    float g0 = __builtin_fabsf(dx * fc + dy * fs) - (hl + oh.x * cd + oh.y * sd);
                    float g1 = __builtin_fabsf(dy * fc - dx * fs) - (hw + oh.x * sd + oh.y * cd);
                    float g2 = __builtin_fabsf(dx * os.y + dy * os.x) - (oh.x + hl * cd + hw * sd);
                    float g3 = __builtin_fabsf(dy * os.y - dx * os.x) - (oh.y + hl * sd + hw * cd);
                    float gap = __builtin_fmaxf(__builtin_fmaxf(g0, g1), __builtin_fmaxf(g2, g3));
                    // fp32 rounding of the centres + trig_eps: the hardware sin/cos error on every product
                    float eps = 1e-3f + 1.9073486e-6f * (mag + __builtin_fabsf(oc.x) + __builtin_fabsf(oc.y)) + trig_eps;
                    // an absent slot has NaN centres: gap is NaN, neither branch below fires
                    bool unsure = (gap <= eps) && (gap >= -eps);
                    unsure = unsure || (dx == 0.0f && dy == 0.0f); // possibly bit-identical boxes
                    if (unsure) fuzzy[w] |= 1ull << jl;
                    else if (gap < -eps) rows_out[w] |= 1ull << jl;
                }
            }
            any_fuzzy = any_fuzzy || fuzzy[w] != 0;
        }
    }
#pragma unroll
    for (int w = 0; w < WV; ++w) mult_rows[w] = rows_out[w];
    PH(3);
    const bool any_fuzzy_wg = block_vote<WV>(L, 1, any_fuzzy) != 0;
    PH(15);
    if (!any_fuzzy_wg) return; // workgroup-uniform; the rest is the rare exact path

    double A[8];
    {
        double s, c; // fp64 sin/cos of the heading: only here, on the exact path
        const double *Kp = SG_TRIG; // opaque: the coefficients are scalar-loaded here instead of living in VGPRs
        asm volatile("" : "+s"(Kp));
        sg_sincos(pose[3], s, c, (ConstTbl)Kp);
        sg_corners(x, y, s, c, L.boxwl[0][sl], L.boxwl[1][sl], bcx, bcy, A);
    }
    if (WV > 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) L.cor[k][sl] = A[k];
        tile_sync<WV>();
    }
    uint64_t eq[WV];
    bool any_eq = false;
#pragma unroll
    for (int w = 0; w < WV; ++w) {
        eq[w] = 0;
        while (sg_any(fuzzy[w] != 0)) {
            // every lane takes part in the cross-lane reads; idle lanes read their own corners
            const bool act = fuzzy[w] != 0;
            const int jl = act ? __builtin_ctzll(fuzzy[w]) : (slot & 63);
            if (act) fuzzy[w] &= fuzzy[w] - 1;
            const int j = tile0 + w * 64 + jl;
            double B[8];
            bool same = true;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                B[k] = WV > 1 ? L.cor[k][j] : shfl_d(A[k], j);
                same = same && (B[k] == A[k]);
            }
            if (act) {
                if (same) eq[w] |= 1ull << jl;                      // g == g_prime: never listed (utils.py:59)
                else if (sg_quads_intersect(A, B)) rows_out[w] |= 1ull << jl;
            }
        }
        any_eq = any_eq || eq[w] != 0;
        mult_rows[w] = rows_out[w];
    }
    if (block_vote<WV>(L, 2, any_eq)) { // geometry -> LAST entity owning it (state/utils.py:32-40)
        int last = slot;
#pragma unroll
        for (int w = 0; w < WV; ++w)
            if (eq[w]) last = max(last, w * 64 + 63 - __builtin_clzll(eq[w]));
        L.last[sl] = last;
        tile_sync<WV>();
        uint64_t nr[WV];
#pragma unroll
        for (int w = 0; w < WV; ++w) nr[w] = 0;
#pragma unroll
        for (int w = 0; w < WV; ++w) {
            uint64_t tmp = rows_out[w];
            while (tmp) {
                int jl = __builtin_ctzll(tmp);
                tmp &= tmp - 1;
                int o = L.last[tile0 + w * 64 + jl];
#pragma unroll
                for (int v = 0; v < WV; ++v)
                    if ((o >> 6) == v) nr[v] |= 1ull << (o & 63);
            }
        }
#pragma unroll
        for (int w = 0; w < WV; ++w) rows_out[w] = nr[w];
    }
    PH(4);
}

// ------------------------------------------------------------------------------------------------
// BatchReplayEntity.add_entities stage 1 (entity/batch.py:83-109): resample every batch-replay
// trajectory onto its scenario's union grid.  One thread per (grid row, entity slot).
// ------------------------------------------------------------------------------------------------
#ifdef SG_UNIT_MAIN // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ void build_grid_kernel(Params p, const int32_t *row_scen /*[totalN]*/, int64_t row0, int64_t row_end)
{
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t row = row0 + gid / p.EP; // grid rows [row0, row_end): sg_upload launches one range per chunk of the knot copy
    int e = (int)(gid % p.EP);
    if (row >= row_end) return;
    int r = row_scen[row];
    uint32_t idx = (uint32_t)r * p.EP + e;
    const LanePtr st(p.stat + (size_t)(idx >> 6) * ST_COUNT * 64, (idx & 63) * 8u);
    int64_t meta = fld<int64_t>(st, ST_META);
    double out[6] = {0, 0, 0, 0, 0, 0};
    if (e < p.E && (meta & 0xff) == SG_KIND_REPLAY) {
        double tq = p.grid_t[row];
        const double *kn = p.knots + fld<int64_t>(st, ST_KNOT_OFF) * 7;
        int n = (int)(meta >> 32);
        if (n == 1) { // batch.py:85-88: second knot at t + 0.1
            double x_lo = kn[0], x_hi = kn[0] + 1e-1;
            for (int c = 0; c < 6; ++c) {
                double v = kn[1 + c];
                if (tq < x_lo || tq > x_hi) out[c] = v;
                else {
                    // searchsorted_left over [x_lo, x_hi] clipped to 1 -> segment (0, 1)
                    double slope = (v - v) / (x_hi - x_lo);
                    out[c] = slope * (tq - x_lo) + v;
                }
            }
        } else if (tq < kn[0]) {
            for (int c = 0; c < 6; ++c) out[c] = kn[1 + c];
        } else if (tq > kn[(size_t)(n - 1) * 7]) {
            for (int c = 0; c < 6; ++c) out[c] = kn[(size_t)(n - 1) * 7 + 1 + c];
        } else {
            int lo = 0, hi = n;
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (kn[(size_t)mid * 7] < tq) lo = mid + 1; else hi = mid;
            }
            int i1 = lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
            const double *a = kn + (size_t)(i1 - 1) * 7, *b = kn + (size_t)i1 * 7;
            for (int c = 0; c < 6; ++c) {
                double slope = (b[1 + c] - a[1 + c]) / (b[0] - a[0]);
                out[c] = slope * (tq - a[0]) + a[1 + c];
            }
        }
    }
    for (int c = 0; c < 6; ++c) p.grid_y[((size_t)row * 6 + c) * p.EP + e] = out[c];
}
#endif // SG_UNIT_MAIN

// ------------------------------------------------------------------------------------------------
// RSSDistances.__call__ (metrics/rss/callback.py:58-128) on the current state of every scenario, + the flags RSS reads
// (metrics/rss/rss.py:70-104).  One workgroup per scenario, one thread per entity.  Ego frame: x lateral, y longitudinal;
// np.dot of 2-vectors = fma(a1, b1, a0 * b0), norm([u, v]) = sqrt(fma(v, v, u * u)) (probed); the per-entity history list
// is carried as (found, last): an "unsafe_*" entry exists / the latest "lateral" | "longitudinal" entry.  Same operation
// sequence as the oracle's sgo_rss_update.
// ------------------------------------------------------------------------------------------------
__device__ inline double rss_dot2(double a0, double a1, double b0, double b1) { return __builtin_fma(a1, b1, a0 * b0); }
#ifdef SG_ABL_RSS_FASTDIV // experiment builds: what do the IEEE divisions of the callback cost (results are wrong)
#define RSS_DIV(a, b) ((a) * __builtin_amdgcn_rcp(b))
#else
#define RSS_DIV(a, b) ((a) / (b))
#endif
__device__ inline void rss_inv_dir(double v0, double v1, double &o0, double &o1)
{
    const double n = sg_norm2(v1, v0);
    o0 = RSS_DIV(v1, n);
    o1 = RSS_DIV(-v0, n);
}
__device__ inline bool rss_on_segment(double ax, double ay, double bx, double by, double px, double py)
{
    return px >= __builtin_fmin(ax, bx) && px <= __builtin_fmax(ax, bx) && py >= __builtin_fmin(ay, by) && py <= __builtin_fmax(ay, by);
}
__device__ inline bool rss_point_in_quad(const double *Q, double px, double py)
{
    bool pos = false, neg = false;
    for (int k = 0; k < 4; ++k) {
        const int m = (k + 1) & 3, o = rn_orient_sign(Q[2 * k], Q[2 * k + 1], Q[2 * m], Q[2 * m + 1], px, py);
        pos |= o > 0;
        neg |= o < 0;
    }
    return !(pos && neg);
}
__device__ inline bool rss_seg_quad(const double *Q, double ax, double ay, double bx, double by)
{
    { // disjoint bounding boxes cannot meet (exact comparisons): the common case, most entities are nowhere near the lines
        const double qx0 = __builtin_fmin(__builtin_fmin(Q[0], Q[2]), __builtin_fmin(Q[4], Q[6]));
        const double qx1 = __builtin_fmax(__builtin_fmax(Q[0], Q[2]), __builtin_fmax(Q[4], Q[6]));
        const double qy0 = __builtin_fmin(__builtin_fmin(Q[1], Q[3]), __builtin_fmin(Q[5], Q[7]));
        const double qy1 = __builtin_fmax(__builtin_fmax(Q[1], Q[3]), __builtin_fmax(Q[5], Q[7]));
        if (qx1 < __builtin_fmin(ax, bx) || qx0 > __builtin_fmax(ax, bx) || qy1 < __builtin_fmin(ay, by) || qy0 > __builtin_fmax(ay, by))
            return false;
    }
    if (rss_point_in_quad(Q, ax, ay) || rss_point_in_quad(Q, bx, by)) return true;
    for (int k = 0; k < 4; ++k) {
        const int m = (k + 1) & 3;
        const double cx = Q[2 * k], cy = Q[2 * k + 1], dx = Q[2 * m], dy = Q[2 * m + 1];
        const int o1 = rn_orient_sign(ax, ay, bx, by, cx, cy), o2 = rn_orient_sign(ax, ay, bx, by, dx, dy);
        const int o3 = rn_orient_sign(cx, cy, dx, dy, ax, ay), o4 = rn_orient_sign(cx, cy, dx, dy, bx, by);
        if (o1 * o2 < 0 && o3 * o4 < 0) return true;
        if ((o1 == 0 && rss_on_segment(ax, ay, bx, by, cx, cy)) || (o2 == 0 && rss_on_segment(ax, ay, bx, by, dx, dy)) ||
            (o3 == 0 && rss_on_segment(cx, cy, dx, dy, ax, ay)) || (o4 == 0 && rss_on_segment(cx, cy, dx, dy, bx, by)))
            return true;
    }
    return false;
}

// RSSDistances for ONE entity against the ego (both present, t != 0): safe distances, the record appended to the entity's
// history, the updated (found | last << 8) state.  Shared by rss_kernel (one update per call) and the rollout variant that
// runs the callback after every step itself.
// DEFER (the rollout variant): the line tests are not run here.  cd = RSS_CD_ISECT: the entity entered the buffer, the
// caller picks unsafe_lateral / unsafe_longitudinal from the history (`last`, else `ab`); otherwise `need` has bit L set for
// every line L whose bounding box meets the entity's (0: cd = 0 is final) and Q is the entity's box in the ego frame.
constexpr int RSS_CD_ISECT = -3;
// The ego's half of one update: its heading and velocity in its own frame and the two inverse directions -- the same for
// every entity of the scenario (callback.py:80-100; four IEEE divisions and three square roots).  (Round 3 moved it to the
// controller pre-pass, once per ego and step, table planes 3-5: -3 % in the rollout kernel, more than that lost to the heavier
// pre-pass beside it -- HISTORY.md.)
struct RssEgo { double eh0, eh1, ei0, ei1, head0, head1, i0, i1, vnorm, vhead, pos1; };
__device__ inline void rss_ego_chain(double es, double ec, double ego_vx, double ego_vy, double ex, double ey, RssEgo &o)
{
    o.eh0 = ec; o.eh1 = es;
    rss_inv_dir(o.eh0, o.eh1, o.ei0, o.ei1);
    o.head0 = rss_dot2(o.eh0, o.eh1, o.ei0, o.ei1);
    o.head1 = rss_dot2(o.eh0, o.eh1, o.eh0, o.eh1);
    const double ego_vel0 = rss_dot2(ego_vx, ego_vy, o.ei0, o.ei1), ego_vel1 = rss_dot2(ego_vx, ego_vy, o.eh0, o.eh1);
    o.pos1 = rss_dot2(ex - ex, ey - ey, o.eh0, o.eh1);
    rss_inv_dir(o.head0, o.head1, o.i0, o.i1);
    o.vnorm = sg_norm2(ego_vel0, ego_vel1);
    o.vhead = rss_dot2(ego_vel0, ego_vel1, o.head0, o.head1);
}
template <bool DEFER = false>
__device__ inline void rss_entity(double ex, double ey, double ego_heading, double ego_vx, double ego_vy, double ego_w, double ego_l,
                                  double hx, double hy, double hh, double hvx, double hvy, double bw, double bl, double bcx,
                                  double bcy, int32_t &state, int &cd, double &s_lat, double &s_long, int *need = nullptr,
                                  double *Qd = nullptr, bool *ab = nullptr, const double *trig = nullptr /* DEFER: sin, cos of
                                  the ego's and of the entity's heading (sg_sincos), computed by the caller */)
{
        const double RESPONSE_TIME = 0.6, MIN_LONG_ACCEL = 1.2 * 9.81, MAX_LONG_ACCEL = 1.2 * 9.81, MIN_SAFE_CLEARANCE = 0.1;
        RssEgo eg;
        {
            double es, ec;
            if (DEFER) { es = trig[0]; ec = trig[1]; }
            else sg_sincos(ego_heading, es, ec);
            rss_ego_chain(es, ec, ego_vx, ego_vy, ex, ey, eg);
        }
        const double eh0 = eg.eh0, eh1 = eg.eh1, ei0 = eg.ei0, ei1 = eg.ei1;
        const double ego_head0 = eg.head0, ego_head1 = eg.head1, ego_pos1 = eg.pos1;
        double hs, hc;
        if (DEFER) { hs = trig[2]; hc = trig[3]; }
        else sg_sincos(hh, hs, hc);
        const double pos0 = rss_dot2(hx - ex, hy - ey, ei0, ei1), pos1 = rss_dot2(hx - ex, hy - ey, eh0, eh1);
        const double head0 = rss_dot2(hc, hs, ei0, ei1), head1 = rss_dot2(hc, hs, eh0, eh1);
        const double vel0 = rss_dot2(hvx, hvy, ei0, ei1), vel1 = rss_dot2(hvx, hvy, eh0, eh1);
        double cor[8], Q[8];
        sg_corners(hx, hy, hs, hc, bw, bl, bcx, bcy, cor);
        for (int k = 0; k < 4; ++k) {
            Q[2 * k] = rss_dot2(cor[2 * k] - ex, cor[2 * k + 1] - ey, ei0, ei1);
            Q[2 * k + 1] = rss_dot2(cor[2 * k] - ex, cor[2 * k + 1] - ey, eh0, eh1);
        }
        { // safe_longitudinal_distance, :231-272
            const double dd = rss_dot2(ego_head0, ego_head1, head0, head1);
            const double m = __builtin_fabs(MAX_LONG_ACCEL * dd), rt = RESPONSE_TIME;
            if (dd > 0) {
                double vf, vr;
                if (ego_pos1 > pos1) { vf = eg.vnorm; vr = rss_dot2(vel0, vel1, ego_head0, ego_head1); }
                else { vf = rss_dot2(vel0, vel1, ego_head0, ego_head1); vr = eg.vnorm; }
                if (vr == 0.0) s_long = MIN_SAFE_CLEARANCE + 0.5 * ego_l;
                else {
                    const double a = vr * rt + __builtin_fmin(RSS_DIV(vf * vf, 2 * m), 0.5 * m * (rt * rt)) +
                                     RSS_DIV((vr + rt * m) * (vr + rt * m), 2 * MIN_LONG_ACCEL) - RSS_DIV(vf * vf, 2 * m);
                    s_long = __builtin_fmax(0.0, a) + MIN_SAFE_CLEARANCE + 0.5 * ego_l;
                }
            } else {
                const double v1 = __builtin_fabs(eg.vhead);
                const double av2 = __builtin_fabs(-__builtin_fabs(rss_dot2(vel0, vel1, ego_head0, ego_head1)));
                const int sp = (pos1 > 0) - (pos1 < 0), sv = (vel1 > 0) - (vel1 < 0);
                if (sp == sv) s_long = MIN_SAFE_CLEARANCE + 0.5 * ego_l;
                else {
                    const double a = (2 * v1 + rt * m) * rt / 2 + RSS_DIV((v1 + rt * m) * (v1 + rt * m), 2 * MIN_LONG_ACCEL) +
                                     (2 * av2 + rt * m) * rt / 2 + RSS_DIV((av2 + rt * m) * (av2 + rt * m), 2 * MIN_LONG_ACCEL);
                    s_long = __builtin_fmax(0.0, a) + MIN_SAFE_CLEARANCE + 0.5 * ego_l;
                }
            }
            s_long = __builtin_fabs(s_long);
        }
        { // safe_lateral_distance, :274-305
            double v = vel0;
            const double i0 = eg.i0, i1 = eg.i1;
            const double ad = __builtin_fabs(rss_dot2(i0, i1, head0, head1));
            const double max_lat = MAX_LONG_ACCEL * ad, min_lat = MIN_LONG_ACCEL * ad, rt = RESPONSE_TIME;
            const int sp = (-pos0 > 0) - (-pos0 < 0), sv = (v > 0) - (v < 0);
            double d0 = 0;
            bool parallel = false;
            if (sp == sv) {
                v = __builtin_fabs(v);
                if (v == 0.0) parallel = true;
                else
                    d0 = __builtin_fmax(0.0, 0.5 * rt * (2 * v + rt * max_lat) + RSS_DIV((v + rt * max_lat) * (v + rt * max_lat), 2 * min_lat) -
                                                 0.5 * (rt * rt) * max_lat - RSS_DIV((rt * max_lat) * (rt * max_lat), 2 * min_lat));
            }
            s_lat = __builtin_fabs(parallel ? MIN_SAFE_CLEARANCE + 0.5 * ego_w : d0 + MIN_SAFE_CLEARANCE + 0.5 * ego_w);
        }
        // unsafe_distance, :179-229
        const int found = state & 0xff, last = (state >> 8) & 0xff;
        if (found) {
            cd = 6;
        } else if (DEFER) {
            const double B[8] = {s_lat, s_long, -s_lat, s_long, -s_lat, -s_long, s_lat, -s_long};
            const double qx0 = __builtin_fmin(__builtin_fmin(Q[0], Q[2]), __builtin_fmin(Q[4], Q[6]));
            const double qx1 = __builtin_fmax(__builtin_fmax(Q[0], Q[2]), __builtin_fmax(Q[4], Q[6]));
            const double qy0 = __builtin_fmin(__builtin_fmin(Q[1], Q[3]), __builtin_fmin(Q[5], Q[7]));
            const double qy1 = __builtin_fmax(__builtin_fmax(Q[1], Q[3]), __builtin_fmax(Q[5], Q[7]));
            // A box strictly beside / above / below the (axis-parallel) buffer is separated by that edge of the buffer in
            // sg_sat_pass(B, Q) too: with finite coordinates the cross products there are +-2 s * (q - +-s), signs exact.
            const double INF = __builtin_inf();
            const bool apart = (qx1 < -s_lat || qx0 > s_lat || qy1 < -s_long || qy0 > s_long) && qx0 > -INF && qx1 < INF &&
                               qy0 > -INF && qy1 < INF && s_lat < INF && s_long < INF;
#ifdef SG_ABL_RSS_NO_SAT
            if (false) {
#else
            if (!apart && sg_quads_intersect(Q, B)) {
#endif
                double j0, j1;
                rss_inv_dir(ego_w, ego_l, j0, j1);
                const double A = __builtin_fabs(__builtin_fabs(pos0) - __builtin_fabs(rss_dot2(pos0, pos1, ego_w, ego_l))) / s_lat;
                const double Bv = __builtin_fabs(__builtin_fabs(pos1 - rss_dot2(pos0, pos1, j0, j1)) / s_long);
                *ab = A > Bv;
                cd = RSS_CD_ISECT;
            } else { // the bounding-box test rss_seg_quad starts with: the two "width" lines (0, 1) are the diagonals of one
                // box, the two "length" lines (2, 3) are horizontal, y = s_long and y = -s_long
                const double lx = 100 * s_lat, ly = 100 * s_long;
                const bool lat_far = qx1 < __builtin_fmin(s_lat, -s_lat) || qx0 > __builtin_fmax(s_lat, -s_lat) ||
                                     qy1 < __builtin_fmin(ly, -ly) || qy0 > __builtin_fmax(ly, -ly);
                const bool long_x_far = qx1 < __builtin_fmin(lx, -lx) || qx0 > __builtin_fmax(lx, -lx);
                const bool far2 = long_x_far || qy1 < __builtin_fmin(s_long, s_long) || qy0 > __builtin_fmax(s_long, s_long);
                const bool far3 = long_x_far || qy1 < __builtin_fmin(-s_long, -s_long) || qy0 > __builtin_fmax(-s_long, -s_long);
                *need = (lat_far ? 0 : 3) | (far2 ? 0 : 4) | (far3 ? 0 : 8);
                cd = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) Qd[k] = Q[k];
            }
        } else {
            const double B[8] = {s_lat, s_long, -s_lat, s_long, -s_lat, -s_long, s_lat, -s_long};
            if (sg_quads_intersect(Q, B)) {
                if (last == 1) cd = 5;
                else if (last == 2) cd = 4;
                else {
                    double j0, j1;
                    rss_inv_dir(ego_w, ego_l, j0, j1);
                    const double A = __builtin_fabs(__builtin_fabs(pos0) - __builtin_fabs(rss_dot2(pos0, pos1, ego_w, ego_l))) / s_lat;
                    const double Bv = __builtin_fabs(__builtin_fabs(pos1 - rss_dot2(pos0, pos1, j0, j1)) / s_long);
                    cd = A > Bv ? 5 : 4;
                }
                state = (state & ~0xff) | (cd == 4 ? 1 : 2);
            } else { // write_intersections, :307-340 (the "length" lines are the buffer's stretched diagonals, as built)
                const bool lat_i = rss_seg_quad(Q, B[0], 100 * B[1], B[4], 100 * B[5]) || rss_seg_quad(Q, B[2], 100 * B[3], B[6], 100 * B[7]);
                const bool long_i = rss_seg_quad(Q, 100 * B[0], B[1], 100 * B[2], B[3]) || rss_seg_quad(Q, 100 * B[4], B[5], 100 * B[6], B[7]);
                cd = lat_i && long_i ? 3 : (lat_i ? 1 : (long_i ? 2 : 0));
                if (cd == 1 || cd == 2) state = (state & 0xff) | (cd << 8);
            }
        }
}

// ---- the line tests of the callback inside the rollout kernel, deferred ----
// write_intersections (callback.py:307-340) asks, for an entity outside the buffer, whether its box meets the buffer's two
// "width" and two "length" lines: exact predicates, ~600 instructions per line, needed by a handful of the 64 lanes of a
// wavefront at a step -- and their only effects are the record of THAT step (read back for the latest update only) and the
// `last` entry of the history, which is looked at when the entity enters the buffer, once.  Inside the step loop they cost
// more than everything else together (every wavefront ran them for its few lanes, and their registers pushed the loop's
// state into scratch).  So a lane whose box meets a line's bounding box only appends a GROUP (its box in the ego frame, the
// safe distances, the lines wanted, the ordinal of the update) to its wavefront's queue in global memory (p.rssq: 96 B,
// (steps of the launch + 1) x 64 groups per wavefront: launches are chunked to fit) and goes on; an entity that enters the
// buffer is flagged in its state word.  rss_lines_kernel runs after the launch, one wavefront per queue: it turns the
// groups into (group, line) items, one per lane, runs the test over full wavefronts, folds the results per owner lane as
// max(ordinal << 3 | code) and max(ordinal << 2 | code in {lateral, longitudinal}) and finishes the records -- the code of
// the latest update, `last`, the class of a pending entry (unsafe_lateral / unsafe_longitudinal from `last`, :196-213).
// Same predicates on the same operands as the per-tick kernel: the results are the same bits.
constexpr int RSSQ_REC = 12;                               // doubles per group record: Q[8], s_lat, s_long, (meta | key << 32), pad
constexpr int RSS_ST_PENDING = 1 << 16, RSS_ST_AB = 1 << 17; // state word: entered the buffer in this launch; A > Bv (:208-212)
constexpr int RSSQ_CAP = 64; // groups per round of rss_lines_kernel
struct RssQueue {
    double q[10][RSSQ_CAP];           // Q[8], s_lat, s_long
    int meta[RSSQ_CAP];               // owner lane | need << 8
    unsigned key[RSSQ_CAP];           // ordinal of the update within the launch
    int hits[RSSQ_CAP];               // bit L: line L meets the box (zero between flushes)
    unsigned short item[4 * RSSQ_CAP];
    unsigned lastword[64], stepcd[64];
};
typedef __attribute__((address_space(3))) RssQueue *RssQueueLds;
#ifdef SG_RSS_STATS
static __device__ unsigned long long sg_rss_stats[8]; // experiment builds: flushes, groups, items, passes, updates (per wavefront)
#define RSS_STAT(i, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&sg_rss_stats[i], (unsigned long long)(v)); } while (0)
#else
#define RSS_STAT(i, v) ((void)0)
#endif

__device__ __forceinline__ void rss_flush_body(RssQueueLds q, int n)
{
    const int lane = threadIdx.x & 63;
    int n_items = 0;
    for (int g0 = 0; g0 < n; g0 += 64) {
        const int g = g0 + lane;
        const int need = g < n ? (q->meta[g] >> 8) & 15 : 0;
#pragma unroll
        for (int L = 0; L < 4; ++L) {
            const bool w = (need >> L) & 1;
            const uint64_t b = __ballot(w);
            if (w) q->item[n_items + __builtin_popcountll(b & ((1ull << lane) - 1))] = (unsigned short)(g | L << 8);
            n_items += __builtin_popcountll(b);
        }
    }
    tile_sync<1>();
    RSS_STAT(0, 1); RSS_STAT(1, n); RSS_STAT(2, n_items); RSS_STAT(3, (n_items + 63) / 64);
    for (int i0 = 0; i0 < n_items; i0 += 64) {
        const int i = i0 + lane;
        if (i < n_items) {
            const int it = q->item[i], g = it & 255, L = it >> 8;
            double Q[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) Q[k] = q->q[k][g];
            const double sl = q->q[8][g], sg = q->q[9][g];
            // B = {sl, sg, -sl, sg, -sl, -sg, sl, -sg}; line 0: (B0, 100 B1)-(B4, 100 B5), 1: (B2, 100 B3)-(B6, 100 B7),
            // 2: (100 B0, B1)-(100 B2, B3), 3: (100 B4, B5)-(100 B6, B7)
            const double lx = 100 * sl, ly = 100 * sg;
            double ax, ay, bx, by;
            if (L == 0) { ax = sl; ay = ly; bx = -sl; by = -ly; }
            else if (L == 1) { ax = -sl; ay = ly; bx = sl; by = -ly; }
            else if (L == 2) { ax = lx; ay = sg; bx = -lx; by = sg; }
            else { ax = -lx; ay = -sg; bx = lx; by = -sg; }
#ifdef SG_ABL_RSS_NO_EVAL
            if (Q[0] == 1e300)
#else
            if (rss_seg_quad(Q, ax, ay, bx, by))
#endif
                __hip_atomic_fetch_or(&q->hits[g], 1 << L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
    }
    tile_sync<1>();
    for (int g0 = 0; g0 < n; g0 += 64) {
        const int g = g0 + lane;
        if (g < n) {
            const int bits = q->hits[g];
            q->hits[g] = 0;
            const bool lat_i = bits & 3, long_i = bits & 12;
            const unsigned cd = lat_i && long_i ? 3 : (lat_i ? 1 : (long_i ? 2 : 0));
            const int owner = q->meta[g] & 63;
            const unsigned k = q->key[g];
            __hip_atomic_fetch_max(&q->stepcd[owner], k << 3 | cd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (cd == 1 || cd == 2) __hip_atomic_fetch_max(&q->lastword[owner], k << 2 | cd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
    }
    tile_sync<1>();
}


// ------------------------------------------------------------------------------------------------
// The rollout kernel: ScenarioGym.reset_scenario / step / rollout (scenario_gym.py:217-267).
//   WV == 1: one 64-lane workgroup carries 64/G scenarios of up to G entities each (tiles of G lanes)
//   WV  > 1: one workgroup of WV wavefronts carries ONE scenario of up to 64*WV entities
// do_reset: State.reset first.  force: step done scenarios too (gym.step()); otherwise each scenario
// stops at is_done (gym.rollout()).  PED: pedestrian agents (social force) are compiled in.
//
// Register-resident per lane across the time loop: pose, distance, the knot segment (x_lo, x_hi,
// y_lo[6], slope[6]), the clock, controller state, ego metric accumulators.  Controller parameters
// and box extents live in LDS; there is no global load in a steady-state step.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ Table lane_table(const Params &p, int kind, const ScenStatic &ss, int slot,
                                            const LanePtr &st)
{
    Table T;
    if (kind == SG_KIND_REPLAY) {
        size_t go = (size_t)ss.grid_off;
        T.x = p.grid_t + go; T.xs = 1;
        T.y = p.grid_y + go * 6 * p.EP + slot; T.ys = 6 * p.EP; T.cs = p.EP;
        T.n = ss.grid_n;
    } else if (kind >= SG_KIND_AGENT_REPLAY) {
        const double *kn = p.knots + fld<int64_t>(st, ST_KNOT_OFF) * 7;
        T.x = kn; T.xs = 7; T.y = kn + 1; T.ys = 7; T.cs = 1;
        T.n = (int)(fld<int64_t>(st, ST_META) >> 32);
    } else {
        T.x = nullptr; T.y = nullptr; T.n = 0; T.xs = T.ys = T.cs = 0;
    }
    return T;
}

//
// TAB: the PID / vehicle agents were integrated by control_kernel; their lanes read (x, y, h) per step from
// its table `tab` instead of running the controller with 1 of 64 lanes active.  TAB launches never reset.
// Register budget of the one-wavefront-per-tile entry points (rollout_kernel_tab / _tab_planar): 168 VGPRs, three
// wavefronts per SIMD; the pre-pass takes a wavefront slot of its own (sgym_hip.hip, launch_rollout).
// HAST (TAB only): the batch has controlled lanes, i.e. there is a table to replay; without it the table code is
// compiled out (batches of replay entities only: the C2 shape).
// ROAD: the ego_off_road terminal condition is compiled in (its own entry point, rollout_kernel_road: the other
// variants keep their register budgets).
// RSSV: the RSSDistances callback (rss_entity) runs after the reset and after every step inside the kernel.
// ------------------------------------------------------------------------------------------------
// Time-sliced replay (launch_sliced in sgym_hip.hip): a batch whose lanes are all replay entities / replay agents is a
// pure function of the clock -- pose_j = interpolant(t_j), presence_j = rule(t_j) -- except for three ORDERED sums
// (State.distances, EgoAvgSpeed, the event list) and the step at which a terminal condition first holds.  A small batch
// (BASELINE config 2: 64 wavefronts on a 1024-SIMD chip) therefore cuts the time axis: the clock t_j = t_{j-1} + dt is
// accumulated once (clock_kernel, the same additions as the step loop), slice s of the steps runs in its own wavefronts
// from a warm-up step that rebuilds state a = s * len out of the clock alone, leaves |delta pose| / ego speed / events per
// step, and an ordered pass (replay_fixup_kernel) adds them up in step order; the state of the last executed step is
// materialised by one more launch (mode 1).  Results are bit-identical to the step-by-step kernel; what is NOT produced is
// the state of every intermediate step in memory.
// ------------------------------------------------------------------------------------------------
struct SliceArgs {
    int mode;            // 0: slices (per-step terms go to the slice arrays), 1: the last executed step with the full state stores
    int n_slices, len;   // slice s covers steps (s * len, min((s + 1) * len, n_total)]
    int n_total;         // steps of the call
    const double *tt;    // [n_clocks][n_total + 1] the clocks: tt[c][j] = State.t after j steps of a scenario that starts at t0_c
    const int *clock_of; // [R] the clock of scenario r (scenarios with the same start time share one)
    double *dnorm;       // [n_blocks][n_total + 1][64] |delta pose[:3]| of step j per lane (+0 when the entity has no pose)
    double2 *espeed;     // [n_total + 1][R] ego speed after step j (NaN: the ego has no pose) and 1 - t_prev / t of
                         // EgoAvgSpeed when the ego's previous update was the previous step (else NaN: the fix-up divides)
    int *first_done;     // [R][n_slices] the step of the slice at which the scenario became done (0x7f7f7f7f: none)
    sg_event *ev;        // [R][n_slices][ev_cap] CollisionMetric events of the slice
    int *nev;            // [R][n_slices]
    const int *n_final;  // [R] (mode 1) the scenario's last executed step
    int slice0;          // first slice of this launch (blockIdx.y counts from it): batches with controlled lanes launch their
                         // slices group by group, each group as soon as the controller pre-pass has reached its last step
};

// Chunked crowd rollouts (launch_crowd_chunks in sgym_hip.hip, sgym_walk.hpp): which scenarios a launch of the crowd kernel
// works on and where they stop.  cls == nullptr: every scenario, n_steps steps.
struct WalkSel {
    const int8_t *cls;     // [R] class of the scenario in this chunk (0 = this kernel, 1 / 2 = walk_kernel<1 / 2>)
    const int32_t *target; // [R] steps-since-reset at which the chunk ends
    int want;              // the class this launch serves; -1: every scenario that has not reached its target (and may run)
};

// CROWD (PED only): every entity of the batch is a pedestrian agent (or padding), default head rotation, no road network:
// no knot segment, no vehicle / replay code, crowd_pairs for the neighbour sums (rollout_kernel_crowd, BASELINE config 5).
// SLICE (TAB, one wavefront per tile): one slice of a time-sliced replay, see SliceArgs.  With HAST the controlled lanes
// (PID / vehicle agents) replay a controller table that spans the WHOLE call -- row j - 1 = the lane after step j, written by
// control_kernel launches that run ahead of the slices -- so a slice that starts at step a finds its lanes' poses there
// like everything else it needs in the clock.
template <int G, int WV, bool PED, bool TAB, bool HAST, bool ROAD = false, bool RSSV = false, bool CROWD = false, bool SLICE = false,
          bool PLANAR = false, bool RIDERS = false, bool CTAB = false>
__device__ __forceinline__ void rollout_body(
    const Params &p, double timestep, int n_steps, int do_reset, int force, const double *actions /*[n][R][2]*/,
    const double *tab /*controller table planes*/, const SliceArgs &sa = SliceArgs{},
    const unsigned bx_arg = ~0u /* the 64-slot block (WV == 1) / scenario of this workgroup when it is not bx: TabGroups */,
    const WalkSel &sel = WalkSel{nullptr, nullptr, 0})
{
    const unsigned bx = bx_arg == ~0u ? blockIdx.x : bx_arg;
    static_assert(!SLICE || (TAB && WV == 1 && !PED && !ROAD && !RSSV), "slices: the table variant, one wavefront per tile");
    static_assert(!(PED && TAB), "pedestrian scenarios run their controllers in the rollout kernel");
    static_assert(!CROWD || (PED && G == 64 && !ROAD && !RSSV), "the crowd variant is a pedestrian variant with 64-lane tiles");
    // RIDERS (crowd variant; its own entry point, rollout_kernel_crowd_riders): the batch also has lanes that are NOT pedestrian
    // agents -- replay entities, replay agents, PID / vehicle agents (a car driving through the crowd, recorded pedestrians).
    // None of them ever looks at another entity (batch.py:34-53, agent.py:125-148, controller.py:105-258), so a pre-pass
    // (control_kernel_riders) has put their pose and presence after every step of the chunk into the controller table, and
    // here they only read their row: the crowd kernel stays free of knot segments and vehicle code.
    static_assert(!RIDERS || CROWD, "riders ride the crowd variant");
    // CTAB (variants with in-kernel controllers whose registers are full -- the RSS callback: rollout_kernel_rss_tab): the PID /
    // vehicle agents were integrated by control_kernel, their lanes read x, y, h of the step from the controller table with a
    // vector load, and the controller code (sin / cos, PID, tangent: ~220 instructions per wavefront-step for one active
    // lane in 64) is not compiled into this kernel at all.
    static_assert(!CTAB || (!TAB && !PED && !CROWD), "CTAB: table rows into an in-kernel-controller variant");
    constexpr int NS = 64 * WV;
    __shared__ TileLds<NS, PED, CROWD> lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t voff = lane * 8u;
    // one wavefront = one 64-slot block of the state arrays: wave-uniform block pointers
    const size_t blk = (size_t)bx * WV + wave;
    const double *st_blk = p.stat + blk * (ST_COUNT * 64);
    const LanePtr st(st_blk, voff);
    const LanePtr dy(p.dyn + blk * ((size_t)(SG_F_COLL + WV) * 64), voff);
    // scenario / slot of this lane
    const int gl = bx * NS + tid;
    const int r_raw = WV == 1 ? gl / G : bx;
    const int slot = WV == 1 ? (gl & (G - 1)) : tid;
    const int tile0 = WV == 1 ? (lane & ~(G - 1)) : 0; // first LDS slot of this lane's tile
    const int sl = tid;                                 // this lane's LDS slot
    const bool in_range = r_raw < p.R;
    const uint32_t r = in_range ? r_raw : p.R - 1;
    const ScenStatic &ss = p.sstat[r];
    sg_scenario_state &sd = p.sdyn[r];
    int step_target = 0x7fffffff;
    if (CROWD && !RIDERS && WV > 1 && sel.cls) { // (one scenario per workgroup: uniform)
        const int tg = sel.target[r];
        const bool mine = sel.want >= 0 ? sel.cls[r] == sel.want : (sd.n_steps < tg && (force || !sd.done));
        if (!mine) return;
        step_target = tg;
    }
    const int64_t meta = fld<int64_t>(st, ST_META);
    const int kind = (in_range && slot < p.E) ? (int)(meta & 0xff) : SG_KIND_NONE;
    const bool is_ped_type = ((meta >> 8) & 0xff) == 1;
    const bool is_ego = in_range && slot == ss.ego;
    const double bcx = fld(st, ST_BCX), bcy = fld(st, ST_BCY);
    const double min_t = fld(st, ST_MIN_T), max_t = fld(st, ST_MAX_T);
    const double length = ss.length;
    const bool is_static = (int)(meta >> 32) == 1;
    const bool is_agent = kind >= SG_KIND_AGENT_REPLAY;
    const bool is_replay = kind == SG_KIND_REPLAY;
    const bool replay_always = p.persist || is_static; // BatchReplayEntity keeps persistent / static entities (batch.py:45-52)
    // per-launch LDS tables: box extents, controller parameters; broad-phase reach of this lane =
    // own bounding-circle radius + the largest radius in the tile + slack
    float rad_thr, trig_eps, nbr_thr = 0.0f;
    float rmax_tile = 0.0f;   // REFINE: the largest bounding-circle radius of the tile ...
    bool hetero = false;      // ... and whether some real entity's is less than two thirds of it
    constexpr bool REFINE = CROWD ? RIDERS : PED;
    {
        const double bw = fld(st, ST_BW), bl = fld(st, ST_BL);
        float rad = (float)(0.5 * __builtin_sqrt(bl * bl + bw * bw)) * 1.000001f;
        float off = (float)__builtin_sqrt(bcx * bcx + bcy * bcy) * 1.000001f;
        // (PED) only entities of type Pedestrian can be somebody's social-force neighbour (pedestrian/sensor.py:56-63): the
        // neighbour reach needs THEIR largest centre offset, not the car's that drives through the crowd
        float rmax = rad, omax = off, omax_ped = (PED && is_ped_type) ? off : 0.0f;
#pragma unroll
        for (int o = 1; o < G; o <<= 1) {
            rmax = __builtin_fmaxf(rmax, __shfl_xor(rmax, o, 64));
            omax = __builtin_fmaxf(omax, __shfl_xor(omax, o, 64));
            if (PED) omax_ped = __builtin_fmaxf(omax_ped, __shfl_xor(omax_ped, o, 64));
        }
        if (WV > 1) { // across the workgroup's wavefronts
            float *red = reinterpret_cast<float *>(lds.cor);
            if (lane == 0) { red[wave] = rmax; red[8 + wave] = omax; red[16 + wave] = omax_ped; }
            __syncthreads();
            for (int w = 0; w < WV; ++w) {
                rmax = __builtin_fmaxf(rmax, red[w]); omax = __builtin_fmaxf(omax, red[8 + w]);
                omax_ped = __builtin_fmaxf(omax_ped, red[16 + w]);
            }
            __syncthreads();
        }
        // hardware sin/cos (error d = SG_TRIG32_ERR per value): each centre moves by <= 2 d off, so the reach grows
        // by 2 d (off + omax); in the filter every gap is a sum of (length <= reach) x (trig product, error <= 4 d)
        rad_thr = rad + rmax + 2e-3f + 2.0f * SG_TRIG32_ERR * (off + omax);
        trig_eps = SG_TRIG32_ERR * (12.0f * rad_thr + 4.0f * (off + omax));
        if (REFINE) {
            rmax_tile = rmax;
            const bool small = kind != SG_KIND_NONE && rad * 1.5f < rmax;
            hetero = WV == 1 ? sg_any(small) : (__syncthreads_or(small) != 0);
        }
        lds.half[sl] = make_float2((float)(0.5 * bl), (float)(0.5 * bw));
        lds.boxwl[0][sl] = bw;
        lds.boxwl[1][sl] = bl;
        if (!TAB) {
#pragma unroll
            for (int q = 0; q < (PED ? 4 : 9); ++q) lds.ctrl[q][sl] = fld(st, ST_CTRL + (PED ? SG_C_PED_SPEED_DESIRED : 0) + q);
        }
        if (PED) // PedestrianSensor radius is measured between reference points; centres differ by the box offsets
            nbr_thr = kind == SG_KIND_AGENT_PEDESTRIAN
                          ? (float)fld(st, ST_CTRL + SG_C_PED_RADIUS) * 1.000001f + off + omax_ped + 2e-3f +
                                2.0f * SG_TRIG32_ERR * (off + omax_ped) : 0.0f;
        if (CROWD) { // thresholds of the radius rule (sg_in_radius), per pedestrian
            const double rr = fld(st, ST_CTRL + SG_C_PED_RADIUS), r2 = rr * rr;
            lds.r2hi[sl] = r2 * (1.0 + 1e-9);
            lds.r2lo[sl] = r2 * 0.9975;
        }
    }
    // CROWD: may this wavefront use crowd_pairs at all?  Default head rotation in every lane, a radius and parameters inside
    // the guards of crowd_pair (wave-uniform, fixed for the launch); the per-step guards are voted in tile_collisions.
    bool crowd_static_ok = false;
    CrowdConsts CC{};
    if (CROWD) {
        const double rr = fld(st, ST_CTRL + SG_C_PED_RADIUS), hr = fld(st, ST_CTRL + SG_C_PED_HEAD_ROT);
        crowd_static_ok = sg_all(kind != SG_KIND_AGENT_PEDESTRIAN || (hr == 0.0 && rr > 0.0 && rr < 0x1p20)) &&
                          crowd_params_ok(p.sf) && !p.ped_serial;
        const RecipDiv rs(p.sf.ped_repulse_sigma);
        CC.k2_scale = p.sf.ped_repulse_V / p.sf.ped_repulse_sigma;
        CC.sig_b = rs.b;
        CC.sig_r = rs.r;
        CC.cos_sight = p.sf.cos_sight;
        CC.sight_weight = p.sf.sight_weight;
        CC.k3 = 2 * p.sf.ped_attract_C;
    }
    bool crowd_ok = false; // workgroup-uniform, per step: the guards of crowd_pair hold for every pedestrian of the tile
    // broad-phase cell size: >= every reach in the tile (+5 % so that fp32 cell coordinates stay consistent)
    float cell_inv;
    {
        float tmax = __builtin_fmaxf(rad_thr, nbr_thr);
#pragma unroll
        for (int o = 1; o < G; o <<= 1) tmax = __builtin_fmaxf(tmax, __shfl_xor(tmax, o, 64));
        if (WV > 1) {
            float *red = reinterpret_cast<float *>(lds.cor);
            if (lane == 0) red[wave] = tmax;
            __syncthreads();
            for (int w = 0; w < WV; ++w) tmax = __builtin_fmaxf(tmax, red[w]);
            __syncthreads();
        }
        cell_inv = 1.0f / (1.05f * tmax + 0.05f);
    }
    // pedestrian route (pedestrian/agent.py:43-47)
    const double *wp = nullptr;
    int nwp = 0;
    if (PED && kind == SG_KIND_AGENT_PEDESTRIAN) {
        int64_t rt = fld<int64_t>(st, ST_ROUTE);
        wp = p.routes + (rt & 0xffffffffffffll) * 2;
        nwp = (int)(rt >> 48);
    }

    // PLANAR (table variant, one wavefront per tile; its own entry point, rollout_kernel_tab_planar): every knot of the
    // batch has z = pitch = roll = +0.0 (sg_upload checks the bit patterns).  Those three channels are then +0.0 in every
    // pose, previous pose and velocity the batch ever holds -- absent lanes included, their rows are zeroed by the reset --
    // so the step neither interpolates, subtracts, tests nor stores them, and their 18 registers (pose, segment base and
    // slope) do not exist.
    static_assert(!PLANAR || (TAB && WV == 1 && !SLICE), "planar: the table variant");
    constexpr bool planar = PLANAR;
    // register-resident across the time loop
    double pose[6], dist, t, prev_t;
    double velx = 0.0, vely = 0.0; // current velocity (social force input), PED only
    CtrlState cs;                 // controller state (agent lanes); pedestrians: speed, goal_idx
    double m_avg, m_max, m_t;     // ego metric accumulators (ego lane)
    uint64_t last_row[WV];        // CollisionMetric.last_timestep (ego lane)
    long long noise_pos = 0;      // variates of the scenario's noise stream consumed so far (PED, noise mode 1)
    int n_ev, goal_idx = 0;
    bool present;
    int done, steps;
    uint64_t row[WV], mult_rows[WV], nbr[WV];
    bool dense = false; // broad-phase strategy of the pedestrian variant (workgroup-uniform), see tile_collisions
    // column of this lane in the controller table (TAB): PID / vehicle agents only
    // The table rows are fetched with SCALAR loads, one controlled lane at a time (at most SG_TAB_LANES per
    // wavefront and wavefront of a wide scenario, checked by the host), one step ahead, and moved into the lane's registers at the end of the step.  A vector load inside the loop would share vmcnt with the state stores and make every
    // step wait for the stores of the previous one.
    const int64_t ctl_q = (TAB && HAST) ? fld<int64_t>(st, ST_CTL) : -1;
    const bool tab_lane = TAB && HAST && ctl_q >= 0 && (kind == SG_KIND_AGENT_PID || kind == SG_KIND_AGENT_VEHICLE);
    const size_t tab_lane_stride = (size_t)(p.tab_steps + 1) * CT_W; // doubles per lane
    // RIDERS: this lane's column of the table (plane 0: x, y, h, speed; plane 2: z, pitch, roll, present)
    const bool rider = (RIDERS && kind != SG_KIND_NONE && kind != SG_KIND_AGENT_PEDESTRIAN) ||
                       (CTAB && (kind == SG_KIND_AGENT_PID || kind == SG_KIND_AGENT_VEHICLE));
    const double *rider_row = (RIDERS || CTAB) ? tab + (size_t)(rider ? fld<int64_t>(st, ST_CTL) : 0) * tab_lane_stride : nullptr;
    int last_k = -1;                                                 // last step of this launch the scenario executed
    constexpr int TL = SG_TAB_LANES(G, WV);
    int cl[TL];                       // wave-uniform: the controlled lanes of this wavefront
    const double *cb[TL];             // wave-uniform: their table columns
    double sx[TL], sy[TL], sh[TL];    // wave-uniform: row of the coming step
    if (TAB && HAST) {
        uint64_t cm = __ballot(tab_lane);
#pragma unroll
        for (int j = 0; j < TL; ++j) {
            cl[j] = -1;
            cb[j] = tab;
            sx[j] = sy[j] = sh[j] = 0.0;
            if (cm) {
                const int l = __builtin_ctzll(cm);
                cm &= cm - 1;
                cl[j] = l;
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ctl_q, l);
                cb[j] = tab + (size_t)lo * tab_lane_stride;
            }
        }
    }
    bool cb_hold[TL]; // wave-uniform (SLICE): the lane's slice starts at the reset state: its first row is consumed by round 1
#pragma unroll
    for (int j = 0; j < TL; ++j) cb_hold[j] = false;
    auto tab_issue = [&](bool first = false) { // s_load the next row of every controlled lane (an unused entry re-reads the first row)
#pragma unroll
        for (int j = 0; j < TL; ++j) {
            ConstTbl rowp = (ConstTbl)cb[j];
            sx[j] = rowp[CT_X];
            sy[j] = rowp[CT_Y];
            sh[j] = rowp[CT_H];
            cb[j] += (cl[j] >= 0 && !(SLICE && first && cb_hold[j])) ? CT_W : 0;
        }
    };

    // do_reset: 0 = continue from the stored state, 1 = State.reset for every scenario, 2 = for the scenarios flagged in
    // p.reset_mask only (one environment of a vector of environments starts a new episode).  The collision pass is a
    // wavefront / workgroup collective and runs outside the per-scenario branch.
    const bool rs = !TAB && (do_reset == 1 || (do_reset == 2 && p.reset_mask[r] != 0));
    double vel[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (rs) {
        // ---- State.reset(t0), state.py:106-143 ----
        const double *kn = p.knots + fld<int64_t>(st, ST_KNOT_OFF) * 7;
        const int nk = (int)(meta >> 32);
        t = ss.t0;
        present = false;
#pragma unroll
        for (int c = 0; c < 6; ++c) { pose[c] = 0.0; vel[c] = 0.0; }
        if (kind != SG_KIND_NONE) {
            bool inside = (t >= min_t) && (t <= max_t);
            if (is_static || inside) { own_position_extrap(kn, nk, t, pose); present = true; }
            else if (p.persist) { // extrapolate=(False, False): clamp
                const double *rowp = t < min_t ? kn : kn + (size_t)(nk - 1) * 7;
#pragma unroll
                for (int c = 0; c < 6; ++c) pose[c] = rowp[1 + c];
                present = true;
            }
            if (present && inside) { // Trajectory.velocity_at_t, trajectory.py:243-273
                const double eps = 1e-4;
                double a[6], b[6];
                own_position_extrap(kn, nk, t + eps / 2, a);
                own_position_extrap(kn, nk, t - eps / 2, b);
#pragma unroll
                for (int c = 0; c < 6; ++c) vel[c] = (a[c] - b[c]) / eps;
            }
        }
        prev_t = t - 0.1; // state.py:135
        dist = 0.0;
        done = 0;
        steps = 0;
        velx = vel[0];
        vely = vel[1];
        cs.speed = present ? sg_norm2(vel[0], vel[1]) : 0.0; // controller.py:100-103
        if (kind == SG_KIND_AGENT_PEDESTRIAN) cs.speed = 0.0; // pedestrian/controller.py:21-23
        cs.e_lon_prev = cs.e_lat_prev = cs.e_lon_int = 0.0;   // controller.py:198-203
        goal_idx = 0;                                         // pedestrian/agent.py:38
        m_avg = m_max = present ? sg_norm3(vel[0], vel[1], vel[2]) : __builtin_nan(""); // metrics/trajectory.py:13-17,36-39
        m_t = 0.0;
#pragma unroll
        for (int w = 0; w < WV; ++w) last_row[w] = 0; // metrics/collision.py:64-68
        n_ev = 0;
        noise_pos = 0;
    } else {
        t = sd.t;
        prev_t = sd.prev_t;
        present = fld<uint64_t>(dy, SG_F_PRESENT) != 0;
#pragma unroll
        for (int c = 0; c < 6; ++c) pose[c] = fld(dy, SG_F_POSE + c);
        velx = fld(dy, SG_F_VEL + 0);
        vely = fld(dy, SG_F_VEL + 1);
        dist = fld(dy, SG_F_DIST);
        if (!TAB) { // the table variant never touches the controller state of lanes it does not own
            cs.speed = fld(dy, SG_F_CTRL + 0); cs.e_lon_prev = fld(dy, SG_F_CTRL + 1);
            cs.e_lat_prev = fld(dy, SG_F_CTRL + 2); cs.e_lon_int = fld(dy, SG_F_CTRL + 3);
        }
        goal_idx = PED ? (int)cs.e_lon_prev : 0; // pedestrians keep goal_idx in the second controller row
        m_avg = sd.ego_avg_speed; m_max = sd.ego_max_speed; m_t = sd.avg_t;
#pragma unroll
        for (int w = 0; w < WV; ++w) last_row[w] = w < 4 ? sd.last_row[w & 3] : sd.last_row_hi[w & 3];
        n_ev = sd.n_events;
        noise_pos = PED ? sd.noise_pos : 0;
        done = sd.done;
        steps = sd.n_steps;
#pragma unroll
        for (int w = 0; w < WV; ++w) row[w] = fld<uint64_t>(dy, SG_F_COLL + w);
    }
    // SLICE: the lane has to be in state `a` (after a steps) before its real steps.  a <= 1: the reset state just loaded
    // (a == 1: + the warm-up step); a >= 2: state a - 1 rebuilt from the clock -- time, and the presence of an agent lane
    // (it has its pose from the reset on, or spawns at step 1: scenario_gym.py:240-244); everything else about that state
    // is either recomputed by the warm-up step (pose, presence of replay lanes, collision row) or not used by it.
    int slice_a = 0;
    const int slice_s = SLICE ? (int)blockIdx.y + sa.slice0 : 0;
    if (SLICE) {
        slice_a = sa.mode == 0 ? slice_s * sa.len : sa.n_final[r] - 1;
        n_ev = 0;
        if (slice_a >= 2) {
            const double *clk = sa.tt + (size_t)sa.clock_of[r] * (size_t)(sa.n_total + 1);
            t = clk[slice_a - 1];
            prev_t = clk[slice_a - 2];
            present = is_agent ? (present || min_t >= ss.t0) : true; // (a replay lane's presence is recomputed by the warm-up step)
            steps = slice_a - 1;
#pragma unroll
            for (int w = 0; w < WV; ++w) { last_row[w] = 0; row[w] = 0; }
        }
        n_steps = sa.mode == 0 ? 1 + min(sa.len, sa.n_total - slice_a) : 2;
        if (TAB && HAST) {
            // a controlled lane that spawns (scenario_gym.py:240-244: absent at the reset, min_t >= t0) took all six channels
            // of its trajectory at the clock of step 1 and keeps z / pitch / roll from then on (controller.py:126-131)
            if (slice_a >= 2 && tab_lane && fld<uint64_t>(dy, SG_F_PRESENT) == 0 && min_t >= ss.t0) {
                const double *clk = sa.tt + (size_t)sa.clock_of[r] * (size_t)(sa.n_total + 1);
                Table T1 = lane_table(p, kind, ss, slot, st);
                Segment S1;
                S1.cur = seg_locate(T1, clk[1]);
                seg_load(T1, S1);
                sg_loads_done();
                const double dq1 = clk[1] - S1.x_lo;
                pose[2] = S1.sl[2] * dq1 + S1.ylo[2];
                pose[4] = S1.sl[4] * dq1 + S1.ylo[4];
                pose[5] = S1.sl[5] * dq1 + S1.ylo[5];
            }
            // round k of this launch consumes row slice_a + k - 1 of the lane's table (round 0 is the warm-up step; a lane
            // that starts from the reset state sits it out and holds row 0 for round 1)
#pragma unroll
            for (int j = 0; j < TL; ++j) {
                if (cl[j] >= 0) {
                    const int a_l = __builtin_amdgcn_readlane(slice_a, cl[j]);
                    cb[j] += (size_t)max(a_l - 1, 0) * CT_W;
                    cb_hold[j] = a_l == 0;
                }
            }
        }
        sg_loads_done();
    }
    if (!TAB && (do_reset != 0 || PED)) {
        // collisions of the reset state; pedestrian scenes also need the neighbour candidates (and LDS positions) of the
        // current state when they continue
        uint64_t tmp_rows[WV];
        tile_collisions<G, WV, PED, CROWD, REFINE>(present, pose, velx, vely, (t + timestep) - t, bcx, bcy, rad_thr, trig_eps, nbr_thr, cell_inv,
                                                   is_ped_type, sl, tile0, lds, tmp_rows, mult_rows, nbr, dense, &crowd_ok, nullptr, hetero, rmax_tile);
        if (rs) {
#pragma unroll
            for (int w = 0; w < WV; ++w) row[w] = tmp_rows[w];
        }
    }
    if (rs) {
        if (in_range) {
#pragma unroll
            for (int c = 0; c < 6; ++c) { stf(dy, SG_F_POSE + c, pose[c]); stf(dy, SG_F_VEL + c, vel[c]); }
            stf(dy, SG_F_DIST, dist);
#pragma unroll
            for (int w = 0; w < WV; ++w) stf(dy, SG_F_COLL + w, row[w]);
            stf(dy, SG_F_PRESENT, (uint64_t)present);
            stf(dy, SG_F_FORCE + 0, 0.0);
            stf(dy, SG_F_FORCE + 1, 0.0);
            if (p.rec_cap > 0) {
#pragma unroll
                for (int c = 0; c < 6; ++c)
                    p.rec_pose[(size_t)c * p.R * p.EP + (size_t)r * p.EP + slot] = present ? pose[c] : __builtin_nan("");
            }
            if (slot == 0) {
                sd.rec_rows = p.rec_cap > 0 ? 1 : 0;
                if (p.rec_cap > 0) p.rec_t[r] = t;
            }
            if (is_ego) sd.ego_distance_travelled = __builtin_nan("");
        }
    }

    // ---- RSSDistances inside the kernel (RSSV) ----
    const uint32_t rss_idx = (uint32_t)r * p.EP + slot;
    int32_t rss_st = 0;
    int rss_cd = -1;
    bool rss_touched = false; // this scenario was updated at least once in this launch
    double rss_lat = __builtin_nan(""), rss_long = __builtin_nan("");
    double rss_bw = 0.0, rss_bl = 0.0, rss_ew = 0.0, rss_el = 0.0;
    int rss_gn = 0;          // groups queued by this wavefront (uniform)
    unsigned rss_k = 0;      // ordinal of this lane's latest update within the launch
    const size_t rss_wave = (size_t)bx * WV + wave;
    SG_GLOBAL v2d *const rss_rec = RSSV ? (SG_GLOBAL v2d *)(p.rssq + rss_wave * (size_t)p.rssq_cap * RSSQ_REC) : nullptr;
    if (RSSV) {
        if (!rs && in_range && slot < p.E) rss_st = p.rss_state[rss_idx];
        rss_bw = fld(st, ST_BW);
        rss_bl = fld(st, ST_BL);
        if (WV == 1) { // the ego is slot 0 of the tile (sg_rss_update refuses anything else)
            rss_ew = shfl_d(rss_bw, tile0);
            rss_el = shfl_d(rss_bl, tile0);
        } else {
            if (tid == 0) { lds.cor[0][0] = rss_bw; lds.cor[1][0] = rss_bl; }
            __syncthreads();
            rss_ew = lds.cor[0][0];
            rss_el = lds.cor[1][0];
            __syncthreads();
        }
    }
    // one RSSDistances.__call__ for this lane's entity; upd: its scenario is being updated (it was reset / it stepped)
    auto rss_call = [&](bool upd, double tnow, double vx, double vy) {
        double ex, ey, eh, evx, evy, trig[4];
        bool ego_pres;
        // sin / cos of every lane's own heading: the entity's for its own box -- and, in the ego's lane, the ego's, which
        // every lane of the tile needs: one evaluation instead of two
        sg_sincos(pose[3], trig[2], trig[3]);
        if (WV == 1) {
            ex = shfl_d(pose[0], tile0); ey = shfl_d(pose[1], tile0); eh = shfl_d(pose[3], tile0);
            evx = shfl_d(vx, tile0); evy = shfl_d(vy, tile0);
            trig[0] = shfl_d(trig[2], tile0); trig[1] = shfl_d(trig[3], tile0);
            ego_pres = (__ballot(present) >> tile0) & 1;
        } else {
            if (tid == 0) {
                lds.cor[0][0] = pose[0]; lds.cor[1][0] = pose[1]; lds.cor[2][0] = pose[3];
                lds.cor[3][0] = vx; lds.cor[4][0] = vy; lds.cor[5][0] = present ? 1.0 : 0.0;
                lds.cor[6][0] = trig[2]; lds.cor[7][0] = trig[3];
            }
            __syncthreads();
            ex = lds.cor[0][0]; ey = lds.cor[1][0]; eh = lds.cor[2][0]; evx = lds.cor[3][0]; evy = lds.cor[4][0];
            ego_pres = lds.cor[5][0] != 0.0;
            trig[0] = lds.cor[6][0]; trig[1] = lds.cor[7][0];
            __syncthreads(); // the collision pass of the next step rewrites the scratch
        }
        int need = 0;
        bool ab = false;
        double Qd[8];
        if (upd) {
            rss_touched = true;
            rss_cd = -1;
            rss_lat = rss_long = __builtin_nan("");
            ++rss_k;
            if (!(tnow == 0.0 || !ego_pres || !present || slot == 0 || slot >= p.E)) // callback.py:76-78
                rss_entity<true>(ex, ey, eh, evx, evy, rss_ew, rss_el, pose[0], pose[1], pose[3], vx, vy, rss_bw, rss_bl, bcx, bcy,
                                 rss_st, rss_cd, rss_lat, rss_long, &need, Qd, &ab, trig);
        }
        // line tests: queued for rss_lines_kernel (see RssQueue)
#ifdef SG_ABL_RSS_NO_PUSH
        need = 0;
#endif
        RSS_STAT(4, 1); RSS_STAT(5, __builtin_popcountll(__ballot((need & 3) != 0))); RSS_STAT(6, __builtin_popcountll(__ballot((need & 12) != 0)));
        const uint64_t wants = __ballot(need != 0);
        if (wants) {
            if (need) {
                const int at = rss_gn + __builtin_popcountll(wants & ((1ull << lane) - 1));
                if (at < p.rssq_cap) { // (always: the host sizes the queue for the steps of the launch)
                    SG_GLOBAL v2d *rec = rss_rec + (size_t)at * (RSSQ_REC / 2);
                    rec[0] = v2d{Qd[0], Qd[1]}; rec[1] = v2d{Qd[2], Qd[3]};
                    rec[2] = v2d{Qd[4], Qd[5]}; rec[3] = v2d{Qd[6], Qd[7]};
                    rec[4] = v2d{rss_lat, rss_long};
                    rec[5] = v2d{__longlong_as_double((long long)((uint64_t)(uint32_t)(lane | need << 8) | (uint64_t)rss_k << 32)), 0.0};
                }
                rss_cd = -4 - (int)rss_k; // "the code of update rss_k is with rss_lines_kernel"
            }
            rss_gn += __builtin_popcountll(wants);
        }
        if (upd && rss_cd == RSS_CD_ISECT) // unsafe_distance, callback.py:196-213: the entry exists from now on, its class is pending
            rss_st = (rss_st & 0xff00) | 3 | RSS_ST_PENDING | (ab ? RSS_ST_AB : 0);
    };
    // State.reset ends with update_callbacks(), state.py:138-140 (the table variant is never the reset launch)
    if (RSSV && !CTAB && do_reset != 0) rss_call(rs, t, vel[0], vel[1]);

    Segment S;
    if (!CROWD) { // (a crowd has no replay lanes: its only trajectory lookup is the rare spawn, done on the spot)
        Table T = lane_table(p, kind, ss, slot, st);
        S.cur = seg_locate(T, t);
        seg_load(T, S);
    }

    // The row of the coming step waits in SGPRs (sx, sy, sh); the step selects it into the controlled lane with
    // scalar-source v_cndmask and then issues the loads of the row after it.
    constexpr bool has_tab = TAB && HAST;
    if (has_tab && n_steps > 0) tab_issue(true);

    // Two nested loops over the same step counter.  The inner one is the steady state and only READS the knot
    // segment S; when some lane's clock is about to cross a knot the wavefront drops to the outer loop, which
    // advances that lane's segment and re-enters.  With the conditional update inside a single loop the compiler
    // keeps two copies of S (28 VGPRs) and moves one onto the other on every step.
    int k = 0;
    bool all_done = false;
    bool vel_clean_prev = false; // wave-uniform
    PhaseTimers ptm;
#ifdef SG_PHASE_TIMERS
    ptm.start();
    if (lane == 0 && bx < 1024) p.phase_cycles[16 + bx * 4 + wave] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); // HW_REG_HW_ID
#endif
    sg_loads_done(); // everything loaded so far is in its registers before the first store is issued
    sg_lgkm_done();
    while (k < n_steps && !all_done) {
    if (!CROWD && t + timestep > S.x_hi) { // rare: next knot segment
        // opaque copies keep the table address arithmetic inside this branch (otherwise ~15 invariant
        // 64-bit row addresses are hoisted out of the time loop and held in VGPRs / spilled)
        int kind_o = kind, slot_o = slot;
        LanePtr st_o = st;
        asm volatile("" : "+v"(kind_o), "+v"(slot_o), "+v"(st_o.a[0]));
        Table T = lane_table(p, kind_o, ss, slot_o, st_o);
        seg_advance(T, S, t + timestep);
        sg_loads_done();
    }
    for (; k < n_steps; ++k) {
        // per wavefront and before any workgroup barrier of the step: does a lane need its next segment?
        if (!CROWD && sg_any(t + timestep > S.x_hi)) break;
        // SLICE: round 0 is the warm-up step (state a - 1 -> a, nothing recorded); a lane that starts from the reset state
        // itself (a == 0) sits it out
        const bool warm = SLICE && k == 0;
        const bool run_lane = in_range && (force || !done) && !(SLICE && k == 0 && slice_a == 0) &&
                              (!(CROWD && !RIDERS && WV > 1) || steps < step_target);
        PH(5);
        // (a workgroup of several wavefronts carries ONE scenario: `run` is already uniform, nothing to vote)
        const bool any_run_ = WV == 1 ? sg_any(run_lane || (SLICE && k == 0 && in_range && !done)) : run_lane;
        PH(7);
        if (!any_run_) { all_done = true; break; }
        // A wavefront that carries ONE scenario (64-lane tiles) has the same `done` in every lane, so past the vote every lane
        // runs: said out loud, the `if (run)` blocks and selects below are not lane-divergent code any more (the compiler
        // cannot see that the 64 copies of `done` agree)
        const bool run = (G == 64 && WV == 1 && !SLICE) ? true : run_lane;
        // coefficient table: opaque per step so the scalar loads stay inside the loop (SGPRs for a few
        // dozen instructions instead of VGPRs for the whole kernel); constant address space => s_load
        const double *Kp = SG_TRIG;
        if (!TAB) asm volatile("" : "+s"(Kp));
        ConstTbl K = (ConstTbl)Kp;

        const double next_t = t + timestep; // scenario_gym.py:229
        const double state_dt = t - prev_t; // State.dt, state.py:198-201
        const double dt = next_t - t;       // = State.dt after this step
        // external actions are the only global loads of a steady-state step: issue them first
        double act_a = 0.0, act_s = 0.0;
        if (!TAB && kind == SG_KIND_AGENT_VEHICLE && actions) {
            const double *a = actions + ((size_t)k * p.R + r) * 2;
            act_a = a[0];
            act_s = a[1];
        }
        double np_[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (PLANAR) {
            double dq = next_t - S.x_lo;
            np_[0] = S.sl[0] * dq + S.ylo[0];
            np_[1] = S.sl[1] * dq + S.ylo[1];
            np_[3] = S.sl[3] * dq + S.ylo[3];
        } else if (!CROWD) {
            double dq = next_t - S.x_lo;
#pragma unroll
            for (int c = 0; c < 6; ++c) np_[c] = S.sl[c] * dq + S.ylo[c];
        }

        // ---- new poses: scenario_gym.py:233-245 ----
        bool npres = false;
        double fpx = 0.0, fpy = 0.0; // PedestrianAgent.force
        bool ped_go = false;
        double ped_fx = 0.0, ped_fy = 0.0, ped_vdes = 0.0;
        if (PED) // the social force of every stepping pedestrian of the wavefront (wave-collective)
            ped_force<WV, CROWD>(p, lds, (int)r, sl, tile0, nbr, is_agent && kind == SG_KIND_AGENT_PEDESTRIAN && present && run, pose,
                                 velx, vely, wp, nwp, goal_idx, ped_go, ped_fx, ped_fy, ped_vdes, K, crowd_static_ok && crowd_ok, CC, &ptm);
        // random fluctuations of the speed and the heading (social_force.py:106-108): np.random.normal(loc, scale) is
        // loc + scale * z; z from the scenario's stream of variates -- two per walking pedestrian, in agent order, as the
        // reference draws them from numpy's global generator -- or from the counter-based generator
        double speed_rand = p.sf.bias_lon, heading_rand = p.sf.bias_lat;
        if (PED && p.noise_mode == 1) { // (launch-uniform branch)
            const uint64_t walk = __ballot(ped_go);
            int before, count;
            if (WV == 1) {
                uint64_t m = walk >> tile0;
                if (G < 64) m &= (1ull << (G & 63)) - 1;
                before = __builtin_popcountll(m & ((1ull << (slot & 63)) - 1));
                count = __builtin_popcountll(m);
            } else { // walkers in the wavefronts before this one: through the fourth vote row
                if (lane == 0) lds.vote[3][wave] = __builtin_popcountll(walk);
                __syncthreads();
                before = __builtin_popcountll(walk & ((1ull << lane) - 1));
                count = 0;
#pragma unroll
                for (int w = 0; w < WV; ++w) {
                    const int c = lds.vote[3][w];
                    before += w < wave ? c : 0;
                    count += c;
                }
                __syncthreads(); // (the row is rewritten next step; stream runs are parity runs, not timing runs)
            }
            const long long at = noise_pos + 2 * before;
            if (ped_go) {
                const bool inside = at + 1 < p.noise_len;
                const double *z = p.noise_normals + (size_t)r * (size_t)p.noise_len + (inside ? at : 0);
                speed_rand = p.sf.bias_lon + p.noise_std_lon * (inside ? z[0] : 0.0);
                heading_rand = p.sf.bias_lat + p.noise_std_lat * (inside ? z[1] : 0.0);
                sg_loads_done();
            }
            if (run) noise_pos += 2 * count;
        } else if (PED && p.noise_mode == 2) {
            double z0, z1;
            sg_noise_pair(p.noise_seed, r, (uint32_t)slot, (uint32_t)steps, z0, z1, K);
            speed_rand = p.sf.bias_lon + p.noise_std_lon * z0;
            heading_rand = p.sf.bias_lat + p.noise_std_lat * z1;
        }
        if (TAB) {
            // Straight-line lane masks (the kernel is bound by instruction issue, branches included):
            // BatchReplayEntity.step (batch.py:34-53) for replay lanes; an agent stays once present and spawns at its
            // trajectory start (scenario_gym.py:240-244); controlled lanes take the pre-pass row, z / p / r unchanged
            // (controller.py:126-131).
            const bool in_window = (next_t >= min_t) & (next_t <= max_t);
            const bool np_replay = replay_always | in_window;
            const bool np_agent = present | (min_t >= t);
            npres = (is_replay & np_replay) | (is_agent & np_agent);
            if (has_tab) {
                const bool take = tab_lane & present & run;
#pragma unroll
                for (int j = 0; j < TL; ++j) { // wave-uniform table row into its lane: v_cndmask with scalar sources
                    const bool tj = take & (lane == cl[j]);
                    np_[0] = tj ? sx[j] : np_[0];
                    np_[1] = tj ? sy[j] : np_[1];
                    np_[3] = tj ? sh[j] : np_[3];
                }
                if (!planar) {
                    np_[2] = take ? pose[2] : np_[2];
                    np_[4] = take ? pose[4] : np_[4];
                    np_[5] = take ? pose[5] : np_[5];
                }
                tab_issue(); // row k + 1 (the table has one spare row), consumed by the next step
            }
        } else if (CROWD) {
            if (RIDERS && rider) { // the pre-pass row of this step: pose and presence after it
                const double4 a = *reinterpret_cast<const double4 *>(rider_row + (size_t)k * CT_W);
                const double4 b = *reinterpret_cast<const double4 *>(rider_row + (size_t)k * CT_W + 2 * (size_t)p.n_ctl_pad * tab_lane_stride);
                sg_loads_done();
                np_[0] = a.x; np_[1] = a.y; np_[3] = a.z;
                np_[2] = b.x; np_[4] = b.y; np_[5] = b.z;
                npres = b.w != 0.0;
            } else if (kind == SG_KIND_AGENT_PEDESTRIAN) {
                if (present) {
                    npres = true;
                    if (run)
                        ped_move(p, ped_go, ped_fx, ped_fy, ped_vdes, lds.ctrl[PED ? SG_C_PED_MAX_SPEED - SG_C_PED_SPEED_DESIRED : 0][sl],
                                 pose, state_dt, cs.speed, fpx, fpy, np_, K, speed_rand, heading_rand);
                } else if (min_t >= t) { // scenario_gym.py:240-244: spawn at the trajectory position of next_t (clamped)
                    npres = true;
                    LanePtr st_o = st;
                    asm volatile("" : "+v"(st_o.a[0]));
                    Table T = lane_table(p, SG_KIND_AGENT_PEDESTRIAN, ss, slot, st_o);
                    Segment S2;
                    S2.cur = seg_locate(T, next_t);
                    seg_load(T, S2);
                    sg_loads_done();
                    const double dq = next_t - S2.x_lo;
#pragma unroll
                    for (int c = 0; c < 6; ++c) np_[c] = S2.sl[c] * dq + S2.ylo[c];
                }
            }
        } else if (kind == SG_KIND_REPLAY) { // BatchReplayEntity.step, batch.py:34-53
            npres = p.persist || is_static || (next_t >= min_t && next_t <= max_t);
        } else if (is_agent) {
            if (present && kind == SG_KIND_AGENT_EXTERNAL) {
                // the caller ran agent.step(state) (agent.py:52-57): its pose, or None = NaN (scenario_gym.py:233-239)
                const double *ep = p.ext_pose + ((size_t)r * p.EP + slot) * 6;
                const double e0 = ep[0];
                if (e0 == e0) {
                    npres = true;
#pragma unroll
                    for (int c = 0; c < 6; ++c) np_[c] = ep[c];
                } else if (p.persist) {
                    npres = true;
#pragma unroll
                    for (int c = 0; c < 6; ++c) np_[c] = pose[c];
                }
                sg_loads_done();
            } else if (present) {
                npres = true;
                if (kind != SG_KIND_AGENT_REPLAY && run) {
                    const double tx = np_[0], ty = np_[1];
#pragma unroll
                    for (int c = 0; c < 6; ++c) np_[c] = pose[c];
                    if (CTAB && (kind == SG_KIND_AGENT_PID || kind == SG_KIND_AGENT_VEHICLE)) {
                        // the pre-pass row of this step; z / pitch / roll stay (controller.py:126-131)
                        const double4 a = *reinterpret_cast<const double4 *>(rider_row + (size_t)k * CT_W);
                        sg_loads_done();
                        np_[0] = a.x; np_[1] = a.y; np_[3] = a.z;
                    } else if (!CTAB && (kind == SG_KIND_AGENT_PID || kind == SG_KIND_AGENT_VEHICLE)) {
                        const double bl = lds.boxwl[1][sl];
                        double sin_h, cos_h; // of the current heading
                        sg_sincos(pose[3], sin_h, cos_h, K);
                        // controller parameters: LDS table; pedestrian scenes keep only the pedestrian rows in LDS and
                        // read these (rare lanes there) from the static rows
                        LanePtr st_c = st;
                        auto cp = [&](int q) -> double {
                            return PED ? fld(st_c, ST_CTRL + q) : lds.ctrl[PED ? 0 : q][sl];
                        };
                        if (kind == SG_KIND_AGENT_PID)
                            pid_step(cs, cp, bl, state_dt, dt, tx, ty, sin_h, cos_h, np_, K);
                        else
                            vehicle_step(cs, cp, bl, dt, act_a, act_s, sin_h, cos_h, np_, K);
                    } else if (PED)
                        ped_move(p, ped_go, ped_fx, ped_fy, ped_vdes,
                                 lds.ctrl[PED ? SG_C_PED_MAX_SPEED - SG_C_PED_SPEED_DESIRED : 0][sl], pose, state_dt,
                                 cs.speed, fpx, fpy, np_, K, speed_rand, heading_rand);
                }
            } else if (min_t >= t) { // scenario_gym.py:240-244: spawn at trajectory start
                npres = true;
            }
        }

        // ---- State.update_poses / update_statistics, state.py:203-239 ----
        double d[6];
        if (PLANAR) {
            d[0] = np_[0] - pose[0]; d[1] = np_[1] - pose[1]; d[3] = np_[3] - pose[3];
            d[2] = d[4] = d[5] = 0.0;
        } else if (!CROWD || !(npres && !present)) { // (the crowd variant keeps the if / else form: its registers are full)
#pragma unroll
            for (int c = 0; c < 6; ++c) d[c] = np_[c] - pose[c];
        }
        if (npres && !present) { // newcomer: previous pose from the extrapolated trajectory, state.py:219-222
            // (the rare case overwrites d: as an if / else the two subtractions were merged behind six copies pose -> prev
            // that every step paid)
            double prev[6];
            LanePtr st_o = st;
            asm volatile("" : "+v"(st_o.a[0]));
            own_position_extrap(p.knots + fld<int64_t>(st_o, ST_KNOT_OFF) * 7,
                                (int)(fld<int64_t>(st_o, ST_META) >> 32), t, prev);
#pragma unroll
            for (int c = 0; c < 6; ++c) d[c] = np_[c] - prev[c];
            if (planar) d[2] = d[4] = d[5] = 0.0; // (0 - 0: the extrapolated channels are +0.0 as well)
        }
        double vel[6];
        // z, pitch and roll rarely move.  `flat`: in every lane that commits a pose this step they keep their value
        // (delta +0.0, entity already present).  Then +0 / dt (dt > 0) is +0 -- the three divisions and range checks
        // are skipped -- and the state blocks already hold these pose rows (and, after one flat step, the +0 velocity
        // rows): they are not stored again.  Memory stays the exact step-materialised state; a steady step issues 9
        // row stores instead of 15.
        bool flat;
        {
            RecipDiv rd(dt);
            if (planar) {
                flat = !SLICE && sg_all(dt > 0.0);
            } else {
                const uint32_t zbits = (uint32_t)(__double2hiint(d[2]) | __double2hiint(d[4]) | __double2hiint(d[5])) |
                                       (uint32_t)(__double2loint(d[2]) | __double2loint(d[4]) | __double2loint(d[5]));
                flat = !SLICE && sg_all((!run | !npres | (present & (zbits == 0))) & (dt > 0.0));
            }
            // RecipDiv::safe for three (six) numerators at once: every |d| below 2^961 through one maximum, and each either
            // +0 or at least 2^-959 (NaN fails the second, infinity the first)
            auto lo_ok = [](double a) { return (__builtin_fabs(a) >= 0x1p-959) | (__double_as_longlong(a) == 0); };
            double dmax = __builtin_fmax(__builtin_fmax(__builtin_fabs(d[0]), __builtin_fabs(d[1])), __builtin_fabs(d[3]));
            bool safe = rd.ok & lo_ok(d[0]) & lo_ok(d[1]) & lo_ok(d[3]);
            if (!flat) {
                dmax = __builtin_fmax(__builtin_fmax(dmax, __builtin_fabs(d[2])), __builtin_fmax(__builtin_fabs(d[4]), __builtin_fabs(d[5])));
                safe = safe & lo_ok(d[2]) & lo_ok(d[4]) & lo_ok(d[5]);
            }
            safe = safe & (dmax < 0x1p961);
            if (sg_all(safe)) {
                vel[0] = rd.div(d[0]); vel[1] = rd.div(d[1]); vel[3] = rd.div(d[3]);
                if (flat) {
                    vel[2] = vel[4] = vel[5] = 0.0;
                } else {
                    vel[2] = rd.div(d[2]); vel[4] = rd.div(d[4]); vel[5] = rd.div(d[5]);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 6; ++c) vel[c] = d[c] / dt;
            }
        }

        // commit (lanes of scenarios that are already done keep their state)
        const bool vel_zpr_clean = vel_clean_prev; // did the previous step leave +0 in every stored z/pitch/roll velocity row?
        vel_clean_prev = flat;
        const bool was_present = present;
        if (run) {
            present = npres;
            if (npres) {
                if (PLANAR) {
                    pose[0] = np_[0]; pose[1] = np_[1]; pose[3] = np_[3];
                } else {
#pragma unroll
                    for (int c = 0; c < 6; ++c) pose[c] = np_[c];
                }
                if (!SLICE) dist += PLANAR ? sg_norm2(d[0], d[1]) /* fma(+0, +0, s) == s for s >= +0 */ : sg_norm3(d[0], d[1], d[2]);
                if (PED) { velx = vel[0]; vely = vel[1]; }
            }
            prev_t = t;
            t = next_t;
            ++steps;
            last_k = k;
            if (SLICE && sa.mode == 0 && !warm) { // the terms of the ordered sums of step `steps` (replay_fixup_kernel)
                sa.dnorm[((size_t)blk * (size_t)(sa.n_total + 1) + (size_t)steps) * 64 + lane] = npres ? sg_norm3(d[0], d[1], d[2]) : 0.0;
                if (is_ego) // (prev_t is the clock before this step; steps == 1: the reset left EgoAvgSpeed.t = 0, the fix-up's case)
                    sa.espeed[(size_t)steps * p.R + r] =
                        make_double2(present ? sg_norm3(vel[0], vel[1], vel[2]) : __builtin_nan(""),
                                     (was_present && steps > 1) ? 1.0 - prev_t / t : __builtin_nan(""));
            }
            // ---- step-materialised state (everything except the collision row, see below) ----
#ifndef SG_ABL_NO_STORES
            if (!SLICE || (sa.mode == 1 && !warm)) {
            stf(dy, SG_F_POSE + 0, pose[0]); stf(dy, SG_F_POSE + 1, pose[1]); stf(dy, SG_F_POSE + 3, pose[3]);
            if (!flat) { stf(dy, SG_F_POSE + 2, pose[2]); stf(dy, SG_F_POSE + 4, pose[4]); stf(dy, SG_F_POSE + 5, pose[5]); }
            if (present) {
                stf(dy, SG_F_VEL + 0, vel[0]); stf(dy, SG_F_VEL + 1, vel[1]); stf(dy, SG_F_VEL + 3, vel[3]);
                if (!(flat && vel_zpr_clean)) { // the rows hold +0 since the previous flat step
                    stf(dy, SG_F_VEL + 2, vel[2]); stf(dy, SG_F_VEL + 4, vel[4]); stf(dy, SG_F_VEL + 5, vel[5]);
                }
            }
            stf(dy, SG_F_DIST, dist);
            stf(dy, SG_F_PRESENT, (uint64_t)present);
            }
#else
            if (k == n_steps - 1 || (steps & 1023) == 0) { // timing ablation only: keep the values live
#pragma unroll
                for (int c = 0; c < 6; ++c) { stf(dy, SG_F_POSE + c, pose[c]); stf(dy, SG_F_VEL + c, vel[c]); }
                stf(dy, SG_F_DIST, dist);
                stf(dy, SG_F_PRESENT, (uint64_t)present);
            }
#endif
            if (PED && kind == SG_KIND_AGENT_PEDESTRIAN) {
                stf(dy, SG_F_FORCE + 0, fpx);
                stf(dy, SG_F_FORCE + 1, fpy);
            }
            if (p.rec_cap > 0 && steps < p.rec_cap) {
                int nan_hi = 0x7ff80000;
                if (!CROWD) asm volatile("" : "+s"(nan_hi)); // (keeps the six selects inside this block: hoisted, they cost every
                                                             // step 18 moves; the crowd variant has no register to spare for it)
                const double absent = __hiloint2double(nan_hi, 0);
#pragma unroll
                for (int c = 0; c < 6; ++c)
                    p.rec_pose[((size_t)steps * 6 + c) * p.R * p.EP + (size_t)r * p.EP + slot] = present ? pose[c] : absent;
                if (slot == 0) { p.rec_t[(size_t)steps * p.R + r] = t; sd.rec_rows = steps + 1; }
            }
            // ---- ego metrics, scenario_gym.py:251-252 ----
            if (!SLICE && is_ego && present && !tab_lane) { // a controlled ego's metrics come with its table (control_kernel)
                double speed = sg_norm3(vel[0], vel[1], vel[2]);
                double w = m_t / t; // EgoAvgSpeed._step, metrics/trajectory.py:19-24
                m_avg += (1.0 - w) * (speed - m_avg);
                m_t = t;
                m_max = __builtin_fmax(speed, m_max); // EgoMaxSpeed, :41-44
            }
        }
        PH(1);
        // ---- State.collisions ----
        uint64_t nrow[WV];
#ifdef SG_ABL_NO_COLL
#pragma unroll
        for (int w = 0; w < WV; ++w) nrow[w] = 0;
#else
        tile_collisions<G, WV, PED, CROWD, REFINE>(present, pose, velx, vely, (t + timestep) - t, bcx, bcy, rad_thr, trig_eps, nbr_thr, cell_inv,
                                                   is_ped_type, sl, tile0, lds, nrow, mult_rows, nbr, dense, &crowd_ok, &ptm, hetero, rmax_tile);
#endif
        if (run) {
#pragma unroll
            for (int w = 0; w < WV; ++w) {
                row[w] = nrow[w];
                if (!SLICE || (sa.mode == 1 && !warm)) stf(dy, SG_F_COLL + w, row[w]);
            }
        }

        // ---- check_terminal, state.py:268-270, 397-408 ----
        int ndone = 0;
        if ((p.term_mask & SG_TERM_MAX_LENGTH) && (t + dt > length)) ndone = 1;
        if (p.term_mask & (SG_TERM_COLLISION | SG_TERM_EGO_COLLISION)) {
            bool any_mine = false;
#pragma unroll
            for (int w = 0; w < WV; ++w) any_mine = any_mine || row[w] != 0;
            bool any_tile, ego0;
            if (WV == 1) {
                uint64_t m = __ballot(any_mine) >> tile0;
                if (G < 64) m &= (1ull << (G & 63)) - 1;
                any_tile = m != 0;
                uint64_t row0 = __shfl(row[0], tile0, 64);
                ego0 = ((__ballot(present) >> tile0) & 1) && row0 != 0;
            } else {
                any_tile = __syncthreads_or(any_mine);
                ego0 = __syncthreads_or(tid == 0 && present && any_mine);
            }
            if ((p.term_mask & SG_TERM_COLLISION) && any_tile) ndone = 1;
            if ((p.term_mask & SG_TERM_EGO_COLLISION) && ego0) ndone = 1;
        }
        if (ROAD && (p.term_mask & SG_TERM_EGO_OFF_ROAD)) {
            // TERMINAL_CONDITIONS["ego_off_road"], state.py:401-407: entities[0] (slot 0 of the tile, not Scenario.ego)
            // absent, or its reference point not strictly inside the driveable surface.  Slot 0 looks its cell up; only
            // cells crossed by a polygon boundary run the exact test.
            bool off = false;
            if (sl == tile0 && in_range) {
                off = true;
                if (present && p.road) {
                    const RoadIndex RI = *p.road;
                    off = !(rn_layers_at(RI, RI.net_of_scen[r], SG_LAYER_DRIVEABLE, pose[0], pose[1]) & SG_LAYER_DRIVEABLE);
                }
            }
            bool off_tile;
            if (WV == 1) off_tile = (__ballot(off) >> tile0) & 1;
            else off_tile = __syncthreads_or(off);
            if (off_tile) ndone = 1;
        }
        if (run) done = ndone;
        if (SLICE && sa.mode == 0 && !warm && run && ndone && sl == tile0)
            sa.first_done[(size_t)r * sa.n_slices + slice_s] = steps; // (once: the scenario does not run after this)

        // ---- CollisionMetric._step, metrics/collision.py:70-75 (ego lane only) ----
        uint64_t ev_fresh0 = 0;   // (ego lane) the hazards of this step's new events, first row word
        int ev_base = -1;         // (ego lane) index of the first of them in the event list; -1: none / not representable
        if (run && is_ego && present) {
            if (!TAB) ev_base = n_ev;
#pragma unroll
            for (int w = 0; w < WV; ++w) {
                uint64_t fresh = row[w] & ~last_row[w];
                if (SLICE && (warm || sa.mode == 1)) fresh = 0; // the events of these steps belong to other launches
                if (!TAB && w == 0) ev_fresh0 = fresh;
                while (fresh) {
                    int j = w * 64 + __builtin_ctzll(fresh);
                    fresh &= fresh - 1;
                    int mult = 1;
                    bool aliased = false;
#pragma unroll
                    for (int v = 0; v < WV; ++v) aliased = aliased || mult_rows[v] != row[v];
                    if (aliased) { // aliased geometries are listed once per owner
                        if (!TAB) ev_base = -1;
                        mult = 0;
#pragma unroll
                        for (int v = 0; v < WV; ++v) {
                            uint64_t tmp = mult_rows[v];
                            while (tmp) { int q = __builtin_ctzll(tmp); tmp &= tmp - 1; mult += lds.last[tile0 + v * 64 + q] == j; }
                        }
                    }
                    // catalog type of the other entity (slot j of this scenario)
                    const int oj = (WV == 1 ? tile0 : 0) + j; // slot inside the workgroup's blocks
                    const double *oblk = p.stat + ((size_t)bx * WV + (oj >> 6)) * (ST_COUNT * 64);
                    int64_t ometa = reinterpret_cast<const int64_t *>(oblk)[ST_META * 64 + (oj & 63)];
                    for (int q = 0; q < mult; ++q) {
                        if (n_ev < p.ev_cap) {
                            sg_event *dst = SLICE ? &sa.ev[((size_t)r * sa.n_slices + slice_s) * p.ev_cap + n_ev]
                                                  : &p.events[(size_t)r * p.ev_cap + n_ev];
                            struct { double t; int32_t scenario, other, type, reserved; } head;
                            head.t = t; head.scenario = (int32_t)r; head.other = j;
                            // 5 = non_vehicle; Vehicle hazards (15 here, -1 once unpacked) wait for classify_events_kernel.
                            // The table variant packs the step of this launch above bit 4: the row of the controller
                            // table that holds the ego's pose at the event (event_ego_pose_kernel unpacks it)
                            // (slices: the table spans the call, the row is the step itself)
                            head.type = (((ometa >> 8) & 0xff) == 0 ? (TAB ? 15 : -1) : 5) | (TAB ? (SLICE ? steps : k + 1) << 4 : 0);
                            head.reserved = 0;
                            *reinterpret_cast<decltype(head) *>(dst) = head;
                            if (!TAB) { // (overwritten below when the hazard is a controlled agent; table launches: event_ego_pose_kernel)
                                double *hp = p.ev_hpose + ((size_t)r * p.ev_cap + n_ev) * 3;
                                hp[0] = hp[1] = hp[2] = __builtin_nan("");
                                if (RIDERS) { // a controlled rider as hazard: its pose after this step is a row of the riders' table
                                    const int okind = (int)(ometa & 0xff);
                                    if (okind == SG_KIND_AGENT_PID || okind == SG_KIND_AGENT_VEHICLE) {
                                        const int64_t octl = reinterpret_cast<const int64_t *>(oblk)[ST_CTL * 64 + (oj & 63)];
                                        const double *hrow = tab + (size_t)octl * tab_lane_stride + (size_t)k * CT_W;
                                        hp[0] = hrow[CT_X]; hp[1] = hrow[CT_Y]; hp[2] = hrow[CT_H];
                                    }
                                }
                            }
                            if (!TAB) { // in-kernel controllers: the ego pose of the event goes along.  (Not in the table
                                        // variant, which has no register to spare: its events are classified right
                                        // after the launch, with the ego pose taken from the table row `reserved`.)
                                double *ep = p.ev_pose + ((size_t)r * p.ev_cap + n_ev) * 3;
                                ep[0] = pose[0]; ep[1] = pose[1]; ep[2] = pose[3];
                            }
                        }
                        ++n_ev;
                    }
                }
                last_row[w] = row[w];
            }
        }
        if (!TAB && WV == 1 && p.ev_cap > 0) {
            // A hazard that is itself a controlled agent (PID / vehicle controller) has no trajectory its pose at the event
            // could be re-derived from: it leaves the pose it has right now beside the event (classify_events_kernel).  The
            // ego lane's new-event mask and list position go to the lanes of its tile; rare, one ballot per step otherwise.
            if (sg_any(ev_base >= 0 && ev_fresh0 != 0)) {
                const int ego_lane = tile0 + ss.ego;
                const uint64_t fr = __shfl(ev_fresh0, ego_lane, 64);
                const int base = __shfl(ev_base, ego_lane, 64);
                if (in_range && base >= 0 && ((fr >> slot) & 1) && (kind == SG_KIND_AGENT_PID || kind == SG_KIND_AGENT_VEHICLE)) {
                    const int at = base + __builtin_popcountll(fr & ((1ull << slot) - 1));
                    if (at < p.ev_cap) {
                        double *hp = p.ev_hpose + ((size_t)r * p.ev_cap + at) * 3;
                        hp[0] = pose[0]; hp[1] = pose[1]; hp[2] = pose[3];
                    }
                }
            }
        }
        if (RSSV) rss_call(run, t, vel[0], vel[1]); // State.step ends with update_callbacks(), state.py:165-171
        if (has_tab) sg_lgkm_done();
        PH(5);
    }
    }
#ifdef SG_PHASE_TIMERS
    ptm.flush(p.phase_cycles);
#endif

    if (SLICE) { // the per-scenario results of a sliced replay are written by replay_fixup_kernel
        if (in_range && is_ego) {
            if (sa.mode == 0) sa.nev[(size_t)r * sa.n_slices + slice_s] = n_ev;
            else {
#pragma unroll
                for (int w = 0; w < WV; ++w) (w < 4 ? sd.last_row[w & 3] : sd.last_row_hi[w & 3]) = last_row[w];
            }
        }
        if (TAB && HAST && sa.mode == 1 && in_range && tab_lane) { // controller state after the last executed step
            const double *lr = tab + (size_t)ctl_q * tab_lane_stride + (size_t)(sa.n_final[r] - 1) * CT_W;
            const double *lr1 = lr + (size_t)p.n_ctl_pad * tab_lane_stride; // plane 1
            stf(dy, SG_F_CTRL + 0, lr[CT_SPEED]); stf(dy, SG_F_CTRL + 1, lr1[CT_ELON]);
            stf(dy, SG_F_CTRL + 2, lr1[CT_ELAT]); stf(dy, SG_F_CTRL + 3, lr1[CT_EINT]);
        }
        return;
    }
    if (RSSV && lane == 0) p.rssq_n[rss_wave] = min(rss_gn, p.rssq_cap);
    // ---- write back what lives in registers during the loop ----
    if (in_range) {
        if (PED && kind == SG_KIND_AGENT_PEDESTRIAN) cs.e_lon_prev = (double)goal_idx;
        if (TAB) {
            if (tab_lane && last_k >= 0) { // controller state after the last executed step
                const double *lr = tab + (size_t)ctl_q * tab_lane_stride + (size_t)last_k * CT_W;
                const double *lr1 = lr + (size_t)p.n_ctl_pad * tab_lane_stride; // plane 1
                stf(dy, SG_F_CTRL + 0, lr[CT_SPEED]); stf(dy, SG_F_CTRL + 1, lr1[CT_ELON]);
                stf(dy, SG_F_CTRL + 2, lr1[CT_ELAT]); stf(dy, SG_F_CTRL + 3, lr1[CT_EINT]);
                if (is_ego) { // ego metrics after the last executed step
                    const double *lr2 = lr1 + (size_t)p.n_ctl_pad * tab_lane_stride; // plane 2
                    m_avg = lr2[CT_MAVG]; m_max = lr2[CT_MMAX]; m_t = lr2[CT_MT];
                }
            }
        } else if ((RIDERS || CTAB) && rider) {
            if (last_k >= 0 && (kind == SG_KIND_AGENT_PID || kind == SG_KIND_AGENT_VEHICLE)) { // controller state after the last executed step
                const double *lr = rider_row + (size_t)last_k * CT_W;
                const double *lr1 = lr + (size_t)p.n_ctl_pad * tab_lane_stride; // plane 1
                stf(dy, SG_F_CTRL + 0, lr[CT_SPEED]); stf(dy, SG_F_CTRL + 1, lr1[CT_ELON]);
                stf(dy, SG_F_CTRL + 2, lr1[CT_ELAT]); stf(dy, SG_F_CTRL + 3, lr1[CT_EINT]);
            }
        } else {
            stf(dy, SG_F_CTRL + 0, cs.speed); stf(dy, SG_F_CTRL + 1, cs.e_lon_prev);
            stf(dy, SG_F_CTRL + 2, cs.e_lat_prev); stf(dy, SG_F_CTRL + 3, cs.e_lon_int);
        }
        if (slot == 0) { sd.t = t; sd.prev_t = prev_t; sd.done = done; sd.n_steps = steps; if (PED) sd.noise_pos = noise_pos; }
        if (RSSV) {
            if (slot < p.E) { // (markers in rss_st / rss_cd: rss_lines_kernel finishes them)
                p.rss_state[rss_idx] = rss_st;
                if (rss_touched) { // the records of the latest update
                    p.rss_code[rss_idx] = rss_cd;
                    p.rss_safe[(size_t)rss_idx * 2] = rss_lat;
                    p.rss_safe[(size_t)rss_idx * 2 + 1] = rss_long;
                }
            }
            if (slot == 0 && rss_touched) p.rss_seen[r] = steps;
        }
        if (is_ego) {
            sd.ego_avg_speed = m_avg; sd.ego_max_speed = m_max; sd.avg_t = m_t;
            if (steps > 0 && present) sd.ego_distance_travelled = dist; // EgoDistanceTravelled, :60-62
#pragma unroll
            for (int w = 0; w < WV; ++w) (w < 4 ? sd.last_row[w & 3] : sd.last_row_hi[w & 3]) = last_row[w];
            sd.n_events = n_ev;
        }
    }
}

// The blocks a launch of a table variant works on (launch_rollout): the 64-slot blocks of the batch are cut into groups of
// `gsz` consecutive blocks -- the host uses one group per rollout PIPELINE, two or three of them, each launched chunk after
// chunk on its own stream -- and a launch runs the groups of `active` only; every group reads the controller-table buffer
// (and runs the number of steps) of the chunk of the time axis IT has reached: buffer index = 2 bits per group in `bufof`.
// One group, active = 1: an ordinary launch over all blocks.
struct TabGroups {
    unsigned long long active, bufof[2];
    int gsz, n[4];
    const double *buf[4];
    // the launch's grid holds the active blocks only (a wavefront that starts just to find its group idle costs ~0.1 us of
    // the dispatcher's time, 0.3 ms for a thousand): grid block i is block start0 + i for i < len0, else start1 + (i - len0)
    unsigned start0, len0, start1, len1;
    __device__ __forceinline__ unsigned map(unsigned i) const { return i < len0 ? start0 + i : start1 + (i - len0); }
    __device__ __forceinline__ bool pick(unsigned blk, int &n_steps, const double *&tab) const
    {
        const unsigned g = blk / (unsigned)gsz;
        if (!((active >> g) & 1)) return false;
        const unsigned b = (unsigned)(bufof[g >> 5] >> (2 * (g & 31))) & 3u;
        n_steps = b == 0 ? n[0] : (b == 1 ? n[1] : (b == 2 ? n[2] : n[3]));
        tab = b == 0 ? buf[0] : (b == 1 ? buf[1] : (b == 2 ? buf[2] : buf[3]));
        return true;
    }
};

template <int G, int WV, bool PED, bool TAB>
__global__ __launch_bounds__(64 * WV, PED ? SG_WAVES_PER_SIMD_PED : (TAB ? SG_WAVES_PER_SIMD_TAB : SG_WAVES_PER_SIMD)) void rollout_kernel(
    Params p, double timestep, int n_steps, int do_reset, int force, const double *actions, const double *tab)
{
    // one wavefront per tile: this entry point serves the batches WITHOUT controlled lanes (rollout_kernel_tab the others)
    rollout_body<G, WV, PED, TAB, (TAB && WV > 1)>(p, timestep, n_steps, do_reset, force, actions, tab);
}

// All-pedestrian batches without road networks (BASELINE config 5): see rollout_body, CROWD
template <int WV>
__global__ __launch_bounds__(64 * WV, SG_WAVES_PER_SIMD_PED) void rollout_kernel_crowd(
    Params p, double timestep, int n_steps, int do_reset, int force, const double *actions, const double *tab, WalkSel sel)
{
    rollout_body<64, WV, true, false, false, false, false, true>(p, timestep, n_steps, do_reset, force, actions, tab, SliceArgs{}, ~0u, sel);
}

// ... with riders: lanes of other kinds whose poses come from the pre-pass table (see rollout_body, RIDERS)
template <int WV>
__global__ __launch_bounds__(64 * WV, SG_WAVES_PER_SIMD_PED) void rollout_kernel_crowd_riders(
    Params p, double timestep, int n_steps, int do_reset, int force, const double *actions, const double *tab)
{
    rollout_body<64, WV, true, false, false, false, false, true, false, false, true>(p, timestep, n_steps, do_reset, force, actions, tab);
}

// terminal_conditions with "ego_off_road": controllers in the kernel, road index lookups for slot 0
template <int G, int WV>
__global__ __launch_bounds__(64 * WV, SG_WAVES_PER_SIMD) void rollout_kernel_road(
    Params p, double timestep, int n_steps, int do_reset, int force, const double *actions, const double *tab)
{
    rollout_body<G, WV, false, false, false, true>(p, timestep, n_steps, do_reset, force, actions, tab);
}

// state_callbacks=[RSSDistances()]: controllers and the RSS callback in the kernel, any number of steps per launch
template <int G, int WV>
__global__ __launch_bounds__(64 * WV, SG_WAVES_PER_SIMD) void rollout_kernel_rss(
    Params p, double timestep, int n_steps, int do_reset, int force, const double *actions, const double *tab)
{
    rollout_body<G, WV, false, false, false, false, true>(p, timestep, n_steps, do_reset, force, actions, tab);
}

// ... with the PID / vehicle agents on the controller pre-pass's table (CTAB): one wavefront per tile
template <int G>
__global__ __launch_bounds__(64, SG_WAVES_PER_SIMD) void rollout_kernel_rss_tab(
    Params p, double timestep, int force, TabGroups tg)
{
    int n_steps;
    const double *tab;
    const unsigned blk = tg.map(blockIdx.x);
    if (!tg.pick(blk, n_steps, tab)) return;
    rollout_body<G, 1, false, false, false, false, true, false, false, false, false, true>(p, timestep, n_steps, 0, force, nullptr, tab,
                                                                                             SliceArgs{}, blk);
}

// ... with the ego_off_road terminal condition / with pedestrian agents (RSSDistances treats every entity alike)
template <int G, int WV>
__global__ __launch_bounds__(64 * WV, SG_WAVES_PER_SIMD) void rollout_kernel_rss_road(
    Params p, double timestep, int n_steps, int do_reset, int force, const double *actions, const double *tab)
{
    rollout_body<G, WV, false, false, false, true, true>(p, timestep, n_steps, do_reset, force, actions, tab);
}
template <int G, int WV>
__global__ __launch_bounds__(64 * WV, SG_WAVES_PER_SIMD_PED) void rollout_kernel_rss_ped(
    Params p, double timestep, int n_steps, int do_reset, int force, const double *actions, const double *tab)
{
    rollout_body<G, WV, true, false, false, false, true>(p, timestep, n_steps, do_reset, force, actions, tab);
}

// The table variant with one wavefront per tile (C2 / C3 shapes) under a 168-VGPR cap: three wavefronts per SIMD.  A
// wavefront of this kernel is latency-bound (1024 steps take 1.7 ms with one wavefront per SIMD, 2.3 ms with three), so the
// third one is nearly free: 73.8 -> 91.7 G entity-steps/s on the C3 shape with z / pitch / roll knots, for 64 B of scratch.
// (Rounds 1-2 held it at 192 so that two of its wavefronts and one of control_kernel (<= 128) filled a SIMD's 512 VGPRs;
// the pre-pass now takes a wavefront slot of its own, one launch per chunk: launch_rollout.)
#ifndef SG_TAB_WAVES // (experiment builds: -DSG_TAB_WAVES=2 -DSG_TAB_VGPR=96)
#define SG_TAB_WAVES 3
#define SG_TAB_VGPR 84
#endif
template <int G>
__global__ __launch_bounds__(64, SG_TAB_WAVES) __attribute__((amdgpu_num_vgpr(SG_TAB_VGPR))) void rollout_kernel_tab(
    Params p, double timestep, int force, TabGroups tg)
{
    int n_steps;
    const double *tab;
    const unsigned blk = tg.map(blockIdx.x);
    if (!tg.pick(blk, n_steps, tab)) return;
    rollout_body<G, 1, false, true, true>(p, timestep, n_steps, 0, force, nullptr, tab, SliceArgs{}, blk);
}
// ... for batches whose knots all have z = pitch = roll = +0.0 (PLANAR)
// Three wavefronts per SIMD (168 VGPRs): the kernel issues ~0.73 of the peak with two, ~0.85 with three.  The pre-pass does
// not fit beside three of them (launch_rollout gives it slots of its own: block groups).
#ifndef SG_PLANAR_WAVES // (experiment builds: -DSG_PLANAR_WAVES=2 -DSG_PLANAR_VGPR=96)
#define SG_PLANAR_WAVES 3
#define SG_PLANAR_VGPR 84
#endif
template <int G>
__global__ __launch_bounds__(64, SG_PLANAR_WAVES) __attribute__((amdgpu_num_vgpr(SG_PLANAR_VGPR))) void rollout_kernel_tab_planar(
    Params p, double timestep, int force, TabGroups tg)
{
    int n_steps;
    const double *tab;
    const unsigned blk = tg.map(blockIdx.x);
    if (!tg.pick(blk, n_steps, tab)) return;
    rollout_body<G, 1, false, true, true, false, false, false, false, true>(p, timestep, n_steps, 0, force, nullptr, tab, SliceArgs{}, blk);
}

// One slice of a time-sliced replay (grid.y = slices; SliceArgs), or its last step with the full state stores
template <int G>
__global__ __launch_bounds__(64, SG_WAVES_PER_SIMD_TAB) void rollout_kernel_slice(Params p, double timestep, SliceArgs sa)
{
    rollout_body<G, 1, false, true, false, false, false, false, true>(p, timestep, 0, 0, 0, nullptr, nullptr, sa);
}
// ... of a batch with controlled lanes: `tab` = the controller table of the whole call (p.tab_steps = sa.n_total)
template <int G>
__global__ __launch_bounds__(64, SG_WAVES_PER_SIMD_TAB) void rollout_kernel_slice_tab(Params p, double timestep, SliceArgs sa, const double *tab)
{
    rollout_body<G, 1, false, true, true, false, false, false, true>(p, timestep, 0, 0, 0, nullptr, tab, sa);
}

// The clocks of a sliced replay: tt[c][j] = State.t after j steps = t0_c + dt + dt + ... (scenario_gym.py:229), the
// additions of the step loop itself; scenarios with the same start time share a clock (launch_sliced).  One lane per clock.
#ifdef SG_UNIT_MAIN // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ __launch_bounds__(64) void clock_kernel(const double *t0 /*[n_clocks]*/, int n_clocks, double timestep, int n_total, double *tt)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= n_clocks) return;
    double *row = tt + (size_t)c * (size_t)(n_total + 1);
    double t = t0[c];
    row[0] = t;
    int j = 1;
    for (; j + 15 <= n_total; j += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { t = t + timestep; v[u] = t; }
#pragma unroll
        for (int u = 0; u < 16; ++u) row[j + u] = v[u];
    }
    for (; j <= n_total; ++j) { t = t + timestep; row[j] = t; }
}
#endif // SG_UNIT_MAIN

// n_final[r] = the step at which scenario r became done (the first over its slices), else all n_total steps
#ifdef SG_UNIT_MAIN // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ __launch_bounds__(64) void slice_final_kernel(Params p, SliceArgs sa, int *n_final, int *done_out)
{
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= p.R) return;
    int nf = 0x7f7f7f7f; // "never": what launch_sliced fills first_done with
    for (int s = 0; s < sa.n_slices; ++s) nf = min(nf, sa.first_done[(size_t)r * sa.n_slices + s]);
    done_out[r] = nf != 0x7f7f7f7f;
    n_final[r] = min(nf, sa.n_total);
}
#endif // SG_UNIT_MAIN

// The ordered pass of a sliced replay, after the last step has been materialised.  replay_fixup_kernel: per entity
// State.distances = the |delta pose| terms added up in step order (state.py:237-239), 32 rows of the block in flight.
// replay_scenario_fixup_kernel: one lane per scenario: the EgoAvgSpeed / EgoMaxSpeed recurrences (metrics/trajectory.py:19-24,
// 41-44; an absent ego skips its update) over the ego's speeds and the clock (two contiguous streams, 16 steps in flight),
// the event lists of the slices concatenated in step order (metrics/collision.py:70-75), and the scenario record.
template <int G>
__global__ __launch_bounds__(64) void replay_fixup_kernel(Params p, SliceArgs sa, const int *n_final)
{
    const int lane = threadIdx.x;
    {
        const size_t blk = blockIdx.x;
        const int gl = (int)blk * 64 + lane;
        const int r_raw = gl / G, slot = gl & (G - 1);
        const bool in_range = r_raw < p.R;
        const int r = in_range ? r_raw : p.R - 1;
        const LanePtr dy(p.dyn + blk * ((size_t)(SG_F_COLL + 1) * 64), lane * 8u);
        const int nf = in_range ? n_final[r] : 0;
        int nf_max = nf;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) nf_max = max(nf_max, __shfl_xor(nf_max, o, 64));
        const double *dn = sa.dnorm + (blk * (size_t)(sa.n_total + 1)) * 64 + lane;
        double dist = 0.0; // State.reset: distances 0 (state.py:136)
        // two buffers of 16 rows: the loads of one are in flight while the other is added up (the additions wait for their
        // own buffer only: loads return in order)
        constexpr int NB = 16;
        double va[NB], vb[NB];
        auto fetch = [&](double (&v)[NB], int j0) {
#pragma unroll
            for (int u = 0; u < NB; ++u) v[u] = dn[(size_t)min(j0 + u, sa.n_total) * 64];
        };
        auto add_up = [&](const double (&v)[NB], int j0) {
#pragma unroll
            for (int u = 0; u < NB; ++u) dist += (j0 + u <= nf) ? v[u] : 0.0; // (x + 0.0 == x for x >= +0: straight-line code)
        };
        int j = 1;
        fetch(va, j);
        for (; j <= nf_max; j += 2 * NB) {
            fetch(vb, j + NB);
            add_up(va, j);
            fetch(va, j + 2 * NB);
            add_up(vb, j + NB);
        }
        if (in_range && slot < p.E) {
            stf(dy, SG_F_DIST, dist);
            if (slot == p.sstat[r].ego && nf > 0 && fld<uint64_t>(dy, SG_F_PRESENT) != 0)
                p.sdyn[r].ego_distance_travelled = dist; // EgoDistanceTravelled, metrics/trajectory.py:60-62
        }
    }
}

// (a controlled ego is no different here: the slices leave its speeds like a replay ego's, the pre-pass skips the metrics)
#ifdef SG_UNIT_MAIN // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ __launch_bounds__(64) void replay_scenario_fixup_kernel(Params p, SliceArgs sa, const int *n_final, const int *done_in)
{
    const int lane = threadIdx.x;
    const int r = (int)blockIdx.x * 64 + lane;
    if (r >= p.R) return;
    sg_scenario_state &sd = p.sdyn[r];
    const int nf = n_final[r];
    double m_avg = sd.ego_avg_speed, m_max = sd.ego_max_speed, m_t = sd.avg_t; // the reset values
    const double2 *es = sa.espeed + r; // [step][R]: the 64 scenarios of the wavefront read one row together
    const double *tr = sa.tt + (size_t)sa.clock_of[r] * (size_t)(sa.n_total + 1);
    // EgoAvgSpeed._step: w = t_prev / t; avg += (1 - w) * (speed - avg).  The slices leave 1 - w whenever t_prev is the
    // previous step's clock (the ego had its pose then): the ordered part is three dependent operations per step.
    auto update = [&](double2 e, double t, bool divide) {
        const bool valid = e.x == e.x;
        double c = e.y;
        if (divide) c = (c != c) ? 1.0 - m_t / t : c; // first update, or the ego was absent in between
        const double a = m_avg + c * (e.x - m_avg);
        m_avg = valid ? a : m_avg;
        m_t = valid ? t : m_t;
        m_max = valid ? __builtin_fmax(e.x, m_max) : m_max;
    };
    // batches of 16 steps, the loads of the next batch in flight while this one is worked through
    constexpr int NB = 16;
    double2 spa[NB], spb[NB];
    double tqa[NB], tqb[NB];
    auto fetch = [&](double2 (&sp)[NB], double (&tq)[NB], int q0) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int qq = min(q0 + u, sa.n_total);
            sp[u] = es[(size_t)qq * p.R];
            tq[u] = tr[qq];
        }
    };
    auto work = [&](const double2 (&sp)[NB], const double (&tq)[NB]) {
        bool need = false; // does a step of the batch have to divide?  (one wave-uniform branch per batch)
#pragma unroll
        for (int u = 0; u < NB; ++u) need |= (sp[u].x == sp[u].x) & (sp[u].y != sp[u].y);
        if (sg_any(need)) {
#pragma unroll
            for (int u = 0; u < NB; ++u) update(sp[u], tq[u], true);
        } else {
#pragma unroll
            for (int u = 0; u < NB; ++u) update(sp[u], tq[u], false);
        }
    };
    int q = 1;
    if (nf >= NB) fetch(spa, tqa, q);
    for (; q + 2 * NB - 1 <= nf; q += 2 * NB) {
        fetch(spb, tqb, q + NB);
        work(spa, tqa);
        fetch(spa, tqa, q + 2 * NB);
        work(spb, tqb);
    }
    if (q + NB - 1 <= nf) { work(spa, tqa); q += NB; }
    for (; q <= nf; ++q) update(es[(size_t)q * p.R], tr[q], true);
    sd.ego_avg_speed = m_avg; sd.ego_max_speed = m_max; sd.avg_t = m_t;
    // events: the slices that lie before the last executed step, in order
    int n_ev = 0;
    for (int s = 0; s < sa.n_slices && s * sa.len < nf; ++s) {
        const int cnt = sa.nev[(size_t)r * sa.n_slices + s];
        const sg_event *src = sa.ev + ((size_t)r * sa.n_slices + s) * p.ev_cap;
        for (int i = 0; i < min(cnt, p.ev_cap); ++i)
            if (n_ev + i < p.ev_cap) p.events[(size_t)r * p.ev_cap + n_ev + i] = src[i];
        n_ev += cnt;
    }
    sd.n_events = n_ev;
    sd.t = tr[nf];
    sd.prev_t = nf > 0 ? tr[nf - 1] : sd.prev_t;
    sd.done = done_in[r];
    sd.n_steps = nf;
}
#endif // SG_UNIT_MAIN

// ------------------------------------------------------------------------------------------------
// Controller pre-pass.  A PIDAgent / external-action VehicleController lane never looks at another
// entity (agent.py:131-148, controller.py:105-140, 205-258: own trajectory, own pose, own controller
// state), so the controlled lanes of the whole batch are gathered 64 to a wavefront and integrated
// here for a chunk of steps; rollout_kernel<.., TAB> then replays the table.  Inside rollout_kernel
// the same work would occupy a full wavefront instruction stream for 1 active lane in 64.
// The step arithmetic is the rollout kernel's own (same device functions, same clock recurrence), so
// both paths produce identical bits.  The lane assumes its scenario keeps running; a scenario that
// terminates early simply stops consuming the table.
//   first: take the lane state from the state blocks (start of an API call); otherwise from
//          p.ctl_state (previous launch of the same call).   k0: step offset into `actions`.
//   row0:  first table row this launch writes (a chunk of the table is filled by several short launches, so
//          that the 64 wavefronts of the pre-pass do not sit on the same SIMDs for a whole chunk).
// ------------------------------------------------------------------------------------------------
// controller parameters and the x / y channels of the knot segment (x_lo, y_lo[2], slope[2]: the PID target) of every
// lane: own column only, no barriers.  With the segment out of the VGPRs (and the LDS under 8 KB per wavefront) the
// kernel compiles for 5 wavefronts per SIMD = 96 VGPRs, which is what fits beside two wavefronts of the rollout kernel.
struct CtlLds { double ctrl[9][64]; double seg[5][64]; };

//   metrics: run the ego's EgoAvgSpeed / EgoMaxSpeed recurrences here (plane 2).  The time-sliced path passes 0: its ordered
//          pass computes them from the speeds the slices leave, and the pre-pass -- a chain of T dependent steps on a handful
//          of wavefronts, the critical path of that mode -- is shorter without them.
// The steady state of a PID lane runs as one straight-line block (`fast` below): every division with a step-invariant or
// shared denominator through a refined reciprocal (RecipDiv: the same bits as `/` inside its operand range), the range
// checks of sin / cos / tan and of the reciprocals as ONE wavefront vote, selects instead of lane branches.  A step in which
// some lane spawns, crosses a knot, saturates its steering beyond the tangent polynomial's range or leaves RecipDiv's range
// runs the general code below it.  Same operations on the same operands in the same order: same bits
// (test_controller_prepass_equals_inline_controllers, SG_CTL_FAST=0 forces the general code).
// FAST: compiled in for control_kernel_fast only (151 VGPRs: the time-sliced path, the RSS table variant and the pipelined
// table path, where the pre-pass chain is the critical path); control_kernel (<= 128 VGPRs) stays as it was.
// RIDERS (control_kernel_riders, for rollout_kernel_crowd_riders): the lanes are ALL non-pedestrian entities of a crowd batch
// -- replay entities (the scenario's union grid, presence rule of batch.py:45-52) and replay agents (own knots, clamped;
// agent.py:125-128) beside the PID / vehicle agents -- and every row also gets plane 2 = z, pitch, roll, presence.
template <bool FAST, bool RIDERS = false>
__device__ __forceinline__ void control_body(const Params &p, double timestep, int n_steps, int first, int k0,
                                             const double *actions /*[n][R][2]*/, double *tab, int row0, int metrics)
{
    __shared__ CtlLds lds;
    const int lane = threadIdx.x;
    const size_t q = (size_t)blockIdx.x * 64 + lane;
    const int ent_raw = p.ctl_ent[q];
    const bool active = ent_raw >= 0;
    const uint32_t ent = active ? (uint32_t)ent_raw : 0u;
    const uint32_t r = ent / (uint32_t)p.EP;
    const LanePtr st(p.stat + (size_t)(ent >> 6) * (ST_COUNT * 64), (ent & 63) * 8u);
    const LanePtr dy(p.dyn + (size_t)(ent >> 6) * ((size_t)p.FROWS * 64), (ent & 63) * 8u);
    const int64_t meta = fld<int64_t>(st, ST_META);
    const int kind = active ? (int)(meta & 0xff) : SG_KIND_NONE;
    const double min_t = fld(st, ST_MIN_T), bl = fld(st, ST_BL);
    // the scenario's ego: its EgoAvgSpeed / EgoMaxSpeed recurrences (metrics/trajectory.py:8-48) run here as well
    const bool is_ego = active && (int)(ent - r * (uint32_t)p.EP) == p.sstat[r].ego;
#pragma unroll
    for (int c = 0; c < 9; ++c) lds.ctrl[c][lane] = fld(st, ST_CTRL + c); // own column only: no barrier needed
    const size_t NP = (size_t)p.n_ctl_pad;
    double *cst = p.ctl_state + q;

    double pose[6], t, prev_t;
    double m_avg, m_max, m_t;
    bool present;
    CtrlState cs;
    if (first) {
        const sg_scenario_state &sd = p.sdyn[r];
        t = sd.t;
        prev_t = sd.prev_t;
        m_avg = sd.ego_avg_speed; m_max = sd.ego_max_speed; m_t = sd.avg_t;
        present = fld<uint64_t>(dy, SG_F_PRESENT) != 0;
#pragma unroll
        for (int c = 0; c < 6; ++c) pose[c] = fld(dy, SG_F_POSE + c);
        cs.speed = fld(dy, SG_F_CTRL + 0); cs.e_lon_prev = fld(dy, SG_F_CTRL + 1);
        cs.e_lat_prev = fld(dy, SG_F_CTRL + 2); cs.e_lon_int = fld(dy, SG_F_CTRL + 3);
    } else {
#pragma unroll
        for (int c = 0; c < 6; ++c) pose[c] = cst[(CS_POSE + c) * NP];
        present = cst[CS_PRESENT * NP] != 0.0;
        cs.speed = cst[(CS_CTRL + 0) * NP]; cs.e_lon_prev = cst[(CS_CTRL + 1) * NP];
        cs.e_lat_prev = cst[(CS_CTRL + 2) * NP]; cs.e_lon_int = cst[(CS_CTRL + 3) * NP];
        t = cst[CS_T * NP];
        prev_t = cst[CS_PREV_T * NP];
        m_avg = cst[(CS_METRIC + 0) * NP]; m_max = cst[(CS_METRIC + 1) * NP]; m_t = cst[(CS_METRIC + 2) * NP];
    }
    if (!active) present = false;

    Table T; // the lane's own knots (PIDAgent target, agent.py:145-148; spawn pose, scenario_gym.py:240-244)
    {
        const double *kn = p.knots + fld<int64_t>(st, ST_KNOT_OFF) * 7;
        T.x = kn; T.xs = 7; T.y = kn + 1; T.ys = 7; T.cs = 1;
        T.n = active ? (int)(meta >> 32) : 0;
    }
    double seg_hi;
    int seg_cur;
    double sgr[5] = {0.0, 0.0, 0.0, 0.0, 0.0}; // FAST: the published segment in registers as well (no LDS read, no wait, per step)
    auto seg_publish = [&](const Segment &S) {
        lds.seg[0][lane] = S.x_lo;
        lds.seg[1][lane] = S.ylo[0]; lds.seg[2][lane] = S.ylo[1];
        lds.seg[3][lane] = S.sl[0]; lds.seg[4][lane] = S.sl[1];
        if (FAST) { sgr[0] = S.x_lo; sgr[1] = S.ylo[0]; sgr[2] = S.ylo[1]; sgr[3] = S.sl[0]; sgr[4] = S.sl[1]; }
        seg_hi = S.x_hi;
        seg_cur = S.cur;
    };
    {
        Segment S;
        S.cur = seg_locate(T, t);
        seg_load(T, S);
        seg_publish(S);
    }
    double *out = tab + (q * (size_t)(p.tab_steps + 1) + (size_t)row0) * CT_W;      // plane 0 rows of this lane
    double *out1 = out + (size_t)p.n_ctl_pad * ((size_t)(p.tab_steps + 1) * CT_W); // plane 1
    double *out2 = out1 + (size_t)p.n_ctl_pad * ((size_t)(p.tab_steps + 1) * CT_W); // plane 2
    // RIDERS: a replay lane's own table (union grid / own knots) and its current segment, all six channels
    const bool replay_lane = RIDERS && (kind == SG_KIND_REPLAY || kind == SG_KIND_AGENT_REPLAY);
    const double max_t = fld(st, ST_MAX_T);
    const bool is_static = (int)(meta >> 32) == 1;
    Table TR{};
    Segment SR{};
    if (RIDERS) {
        TR = lane_table(p, replay_lane ? kind : SG_KIND_NONE, p.sstat[r], (int)(ent - r * (uint32_t)p.EP), st);
        if (!replay_lane) TR.n = 0;
        SR.cur = seg_locate(TR, t);
        seg_load(TR, SR);
    }
    sg_loads_done();
    // launch-invariant part of the fast path's vote; reciprocals of the step-invariant denominators
    const bool fast_kind = FAST && !p.ctl_general && sg_all(!active || (kind == SG_KIND_AGENT_PID && bl > 0.0 && bl < 0x1p400));
    const RecipDiv rd_l(active ? bl : 1.0), rd_10(10.0);
    double cpr[9]; // FAST: the controller parameters in registers
#pragma unroll
    for (int c = 0; c < 9; ++c) cpr[c] = FAST ? fld(st, ST_CTRL + c) : 0.0;
    sg_loads_done();

    for (int k = 0; k < n_steps; ++k) {
        const double *Kp = SG_TRIG;
        if (!FAST) asm volatile("" : "+s"(Kp)); // (FAST has registers to spare: the coefficients may live in them for the whole launch)
        ConstTbl K = (ConstTbl)Kp;
        const double next_t = t + timestep; // the rollout kernel's clock, scenario_gym.py:229
        const double state_dt = t - prev_t;
        const double dt = next_t - t;
        if (FAST && fast_kind && sg_all(!active || (present && !(next_t > seg_hi) && __builtin_fabs(pose[3]) < 1.0e5))) {
            // ---- PIDController._step + VehicleController._step (controller.py:205-258, 105-140), straight line ----
            const double dq = next_t - sgr[0];
            const double tx = sgr[3] * dq + sgr[1], ty = sgr[4] * dq + sgr[2];
            double sin_h, cos_h;
            sg_sincos_core(pose[3], sin_h, cos_h, K);
            const double e0 = tx - pose[0], e1 = ty - pose[1];
            const double e_lon = cos_h * e0 + sin_h * e1;
            const double e_lat = -sin_h * e0 + cos_h * e1;
            const double speed0 = cs.speed;
            const double g_mid = 1.0 - rd_10.div(0.9 * (speed0 - 5.0)); // (speed in (5, 15]: the numerator is in RecipDiv's range)
            const double gain = (speed0 > 5.0 && speed0 <= 15) ? g_mid : (speed0 > 15 ? 0.1 : 1.0);
            const RecipDiv rd(state_dt);
            const double d_lat = e_lat - cs.e_lat_prev, d_lon = e_lon - cs.e_lon_prev;
            const double e_lat_D = rd.div(d_lat);
            const double kp = cpr[SG_C_STEER_KP] * gain, kd = cpr[SG_C_STEER_KD] * gain;
            double steer = kp * e_lat + kd * e_lat_D;
            const double e_lon_D = rd.div(d_lon);
            const double e_lon_I = cs.e_lon_int + e_lon * state_dt;
            double accel = cpr[SG_C_ACCEL_KP] * e_lon + cpr[SG_C_ACCEL_KD] * e_lon_D + cpr[SG_C_ACCEL_KI] * e_lon_I;
            accel = __builtin_fabs(e_lon) > 0.1 ? accel : 0.0;
            const double max_steer = cpr[SG_C_MAX_STEER], max_accel = cpr[SG_C_MAX_ACCEL];
            const double max_speed = cpr[SG_C_MAX_SPEED], allow_rev = cpr[SG_C_ALLOW_REVERSE];
            accel = __builtin_fmin(__builtin_fmax(accel, -max_accel), max_accel);
            steer = __builtin_fmin(__builtin_fmax(steer, -max_steer), max_steer);
            const double dxs = speed0 * cos_h, dys = speed0 * sin_h;
            // tan(steer): the polynomial below 0.67434, sin / cos above (sg_tan); a saturated steering angle is common
            // enough among 64 lanes that both live here, the second under a wave-uniform branch
            double tan_s = sg_tan_poly(steer, K);
            const bool steep = !(__builtin_fabs(steer) < 0.67434);
            if (sg_any(steep & active)) {
                double s2, c2;
                sg_sincos_core(steer, s2, c2, K);
                tan_s = steep ? s2 / c2 : tan_s;
            }
            const double hnum = speed0 * tan_s;
            const double dh = hnum == 0.0 ? hnum : rd_l.div(hnum); // (+-0 / l = +-0 for l > 0)
            const double nx = pose[0] + dxs * dt, ny = pose[1] + dys * dt, nh = pose[3] + dh * dt;
            double nspeed = speed0 + accel * dt;
            nspeed = allow_rev == 0.0 ? __builtin_fmax(0.0, nspeed) : nspeed;
            nspeed = max_speed == max_speed ? __builtin_fmin(max_speed, nspeed) : nspeed;
            // the one vote on everything the straight-line forms assumed
            bool ok = rd.safe(d_lat) & rd.safe(d_lon) & (__builtin_fabs(steer) < 1.0e5) & (rd_l.safe(hnum) | (hnum == 0.0));
            // State.update_statistics for the ego + its metrics (state.py:230-239, metrics/trajectory.py:19-24, 41-44)
            double n_avg = m_avg, n_max = m_max, n_mt = m_t;
            if (metrics) { // (launch-uniform)
                const RecipDiv rdt(dt), rnt(next_t);
                const double ax = nx - pose[0], ay = ny - pose[1];
                const double az = pose[2] - pose[2]; // z stays (controller.py:126-131): +0 unless it is not finite
                const double speed = sg_norm3(rdt.div(ax), rdt.div(ay), 0.0); // (+0 / dt = +0)
                const double w = rnt.div(m_t);
                n_avg = m_avg + (1.0 - w) * (speed - m_avg);
                n_max = __builtin_fmax(speed, m_max);
                n_mt = next_t;
                ok = ok & (!is_ego | (rdt.safe(ax) & rdt.safe(ay) & rnt.safe(m_t) & (dt > 0.0) & (az == 0.0)));
            }
            if (sg_all(!active || ok)) {
                cs.e_lat_prev = e_lat; cs.e_lon_prev = e_lon; cs.e_lon_int = e_lon_I; cs.speed = nspeed;
                pose[0] = nx; pose[1] = ny; pose[3] = nh;
                if (is_ego) { m_avg = n_avg; m_max = n_max; m_t = n_mt; }
                prev_t = t;
                t = next_t;
                *reinterpret_cast<double4 *>(out + (size_t)k * CT_W) = make_double4(pose[0], pose[1], pose[3], cs.speed);
                *reinterpret_cast<double4 *>(out1 + (size_t)k * CT_W) = make_double4(cs.e_lon_prev, cs.e_lat_prev, cs.e_lon_int, 0.0);
                if (is_ego && metrics) *reinterpret_cast<double4 *>(out2 + (size_t)k * CT_W) = make_double4(m_avg, m_max, m_t, 0.0);
                continue;
            }
        }
        double act_a = 0.0, act_s = 0.0;
        if (kind == SG_KIND_AGENT_VEHICLE && actions) {
            const double *a = actions + ((size_t)(k0 + k) * p.R + r) * 2;
            act_a = a[0];
            act_s = a[1];
        }
        if (next_t > seg_hi) {
            Segment S;
            S.x_hi = seg_hi;
            S.cur = seg_cur;
            seg_advance(T, S, next_t);
            seg_publish(S);
            sg_loads_done();
        }
        double np_[6];
        const double dq = next_t - lds.seg[0][lane];
        np_[0] = lds.seg[3][lane] * dq + lds.seg[1][lane]; // PID target (x, y) at next_t
        np_[1] = lds.seg[4][lane] * dq + lds.seg[2][lane];
        bool npres = false;
        if (RIDERS && replay_lane) {
            if (next_t > SR.x_hi) { seg_advance(TR, SR, next_t); sg_loads_done(); }
            const double dqr = next_t - SR.x_lo;
#pragma unroll
            for (int c = 0; c < 6; ++c) np_[c] = SR.sl[c] * dqr + SR.ylo[c];
            npres = kind == SG_KIND_REPLAY ? (p.persist || is_static || (next_t >= min_t && next_t <= max_t))  // batch.py:45-52
                                           : (present || min_t >= t);                                         // scenario_gym.py:233-244
        } else if (present) {
            npres = true;
            const double tx = np_[0], ty = np_[1];
#pragma unroll
            for (int c = 0; c < 6; ++c) np_[c] = pose[c];
            double sin_h, cos_h;
            sg_sincos(pose[3], sin_h, cos_h, K);
            auto cp = [&](int q) -> double { return lds.ctrl[q][lane]; };
            if (kind == SG_KIND_AGENT_PID)
                pid_step(cs, cp, bl, state_dt, dt, tx, ty, sin_h, cos_h, np_, K);
            else
                vehicle_step(cs, cp, bl, dt, act_a, act_s, sin_h, cos_h, np_, K);
        } else if (active && min_t >= t) { // spawn at the trajectory position: all six channels of the bracket
            npres = true;
            Segment S;
            S.cur = seg_cur;
            seg_load(T, S);
            sg_loads_done();
            const double dqs = next_t - S.x_lo;
#pragma unroll
            for (int c = 0; c < 6; ++c) np_[c] = S.sl[c] * dqs + S.ylo[c];
        }
        if (is_ego && npres && metrics) { // State.update_statistics for this lane (state.py:230-239) + the ego metrics
            double prev[6];
            if (!present) { // newcomer: previous pose from the extrapolated trajectory, state.py:219-222
                own_position_extrap(T.x, T.n, t, prev);
                sg_loads_done();
            } else {
#pragma unroll
                for (int c = 0; c < 6; ++c) prev[c] = pose[c];
            }
            const double v0 = (np_[0] - prev[0]) / dt, v1 = (np_[1] - prev[1]) / dt, v2 = (np_[2] - prev[2]) / dt;
            const double speed = sg_norm3(v0, v1, v2);
            const double w = m_t / next_t; // EgoAvgSpeed._step, metrics/trajectory.py:19-24
            m_avg += (1.0 - w) * (speed - m_avg);
            m_t = next_t;
            m_max = __builtin_fmax(speed, m_max); // EgoMaxSpeed, :41-44
        }
        present = npres;
        if (npres) {
#pragma unroll
            for (int c = 0; c < 6; ++c) pose[c] = np_[c];
        }
        prev_t = t;
        t = next_t;
        *reinterpret_cast<double4 *>(out + (size_t)k * CT_W) = make_double4(pose[0], pose[1], pose[3], cs.speed);
        *reinterpret_cast<double4 *>(out1 + (size_t)k * CT_W) = make_double4(cs.e_lon_prev, cs.e_lat_prev, cs.e_lon_int, 0.0);
        if (is_ego && metrics) *reinterpret_cast<double4 *>(out2 + (size_t)k * CT_W) = make_double4(m_avg, m_max, m_t, 0.0);
        if (RIDERS) *reinterpret_cast<double4 *>(out2 + (size_t)k * CT_W) = make_double4(pose[2], pose[4], pose[5], present ? 1.0 : 0.0);
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) cst[(CS_POSE + c) * NP] = pose[c];
    cst[CS_PRESENT * NP] = present ? 1.0 : 0.0;
    cst[(CS_CTRL + 0) * NP] = cs.speed; cst[(CS_CTRL + 1) * NP] = cs.e_lon_prev;
    cst[(CS_CTRL + 2) * NP] = cs.e_lat_prev; cst[(CS_CTRL + 3) * NP] = cs.e_lon_int;
    cst[CS_T * NP] = t;
    cst[CS_PREV_T * NP] = prev_t;
    cst[(CS_METRIC + 0) * NP] = m_avg; cst[(CS_METRIC + 1) * NP] = m_max; cst[(CS_METRIC + 2) * NP] = m_t;
}

#ifdef SG_UNIT_CTL // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ __launch_bounds__(64, SG_CTL_WAVES) void control_kernel(Params p, double timestep, int n_steps, int first, int k0,
                                                     const double *actions /*[n][R][2]*/, double *tab, int row0, int metrics)
{
    control_body<false>(p, timestep, n_steps, first, k0, actions, tab, row0, metrics);
}
#endif // SG_UNIT_CTL
#ifdef SG_UNIT_CTL // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ __launch_bounds__(64, 2) void control_kernel_riders(Params p, double timestep, int n_steps, int first, int k0,
                                                               const double *actions /*[n][R][2]*/, double *tab, int row0, int metrics)
{
    control_body<false, true>(p, timestep, n_steps, first, k0, actions, tab, row0, 0);
}
#endif // SG_UNIT_CTL
#ifdef SG_UNIT_CTL // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ __launch_bounds__(64, 1) void control_kernel_fast(Params p, double timestep, int n_steps, int first, int k0,
                                                             const double *actions /*[n][R][2]*/, double *tab, int row0, int metrics)
{
    control_body<true>(p, timestep, n_steps, first, k0, actions, tab, row0, metrics);
}
#endif // SG_UNIT_CTL

// ------------------------------------------------------------------------------------------------
// FutureCollisionDetector._step (sensor/common.py:87-106), SURVEY 8f N2: does the ego's box, moved along its
// trajectory to n sample times in [t, t + horizon] (np.linspace), overlap any other entity's box at that entity's own
// trajectory position (clamped outside the trajectory; presence is not consulted)?  One workgroup per scenario, one
// thread per entity slot, exact fp64 predicate (the operation sequence of the oracle), geometry equal to the ego's
// never counts (utils.py:59).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void own_position_clamped(const double *kn, int n, double t, double (&out)[6])
{
    // Trajectory.position_at_t with the default extrapolate=(False, False): trajectory.py:185-196
    const double *last = kn + (size_t)(n - 1) * 7;
    if (t < kn[0]) {
#pragma unroll
        for (int c = 0; c < 6; ++c) out[c] = kn[1 + c];
    } else if (t > last[0]) {
#pragma unroll
        for (int c = 0; c < 6; ++c) out[c] = last[1 + c];
    } else {
        own_position_extrap(kn, n, t, out);
    }
}

// One workgroup per scenario; the (entity, sample) pairs are spread over its 256 threads (the binary searches over the
// knots are chains of dependent loads: 10 samples one after the other per entity thread took 250 us for 4096 x 64).
// Pass 1: the ego's corners at every sample time into LDS; pass 2: every other pair against them.
#define SG_FUT_MAX_SAMPLES 64
#ifdef SG_UNIT_MAIN // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ __launch_bounds__(256) void future_kernel(Params p, double horizon, int n_samples, unsigned char *out /*[R]*/)
{
    __shared__ double ego_c[SG_FUT_MAX_SAMPLES][8];
    const int r = blockIdx.x, tid = threadIdx.x;
    const ScenStatic &ss = p.sstat[r];
    const double start = p.sdyn[r].t, stop = start + horizon;
    const double step = n_samples > 1 ? (stop - start) / (double)(n_samples - 1) : 0.0; // np.linspace
    auto corners_at = [&](int e, int j, double *C) -> bool {
        const uint32_t idx = (uint32_t)r * p.EP + e;
        const LanePtr st(p.stat + (size_t)(idx >> 6) * (ST_COUNT * 64), (idx & 63) * 8u);
        const int64_t meta = fld<int64_t>(st, ST_META);
        if ((int)(meta & 0xff) == SG_KIND_NONE) return false;
        double tj = (double)j * step + start;
        if (n_samples > 1 && j == n_samples - 1) tj = stop;
        double pose[6], s, c;
        own_position_clamped(p.knots + fld<int64_t>(st, ST_KNOT_OFF) * 7, (int)(meta >> 32), tj, pose);
        sg_sincos(pose[3], s, c);
        sg_corners(pose[0], pose[1], s, c, fld(st, ST_BW), fld(st, ST_BL), fld(st, ST_BCX), fld(st, ST_BCY), C);
        return true;
    };
    bool hit = false;
    for (int j0 = 0; j0 < n_samples; j0 += SG_FUT_MAX_SAMPLES) { // more samples than the LDS table holds: in rounds
        const int nj = min(SG_FUT_MAX_SAMPLES, n_samples - j0);
        if (tid < nj) {
            double C[8];
            corners_at(ss.ego, j0 + tid, C); // the ego is an entity of the scenario: never SG_KIND_NONE
#pragma unroll
            for (int k = 0; k < 8; ++k) ego_c[tid][k] = C[k];
        }
        __syncthreads();
        for (int w = tid; w < nj * p.E; w += 256) {
            const int j = w / p.E, e = w - j * p.E;
            double C[8];
            if (e == ss.ego || !corners_at(e, j0 + j, C)) continue;
            double A[8];
            bool same = true;
#pragma unroll
            for (int k = 0; k < 8; ++k) { A[k] = ego_c[j][k]; same = same && (A[k] == C[k]); }
            if (!same && sg_quads_intersect(A, C)) hit = true;
        }
        __syncthreads();
    }
    const int any = __syncthreads_or(hit);
    if (tid == 0) out[r] = (unsigned char)(any != 0);
}
#endif // SG_UNIT_MAIN

// ------------------------------------------------------------------------------------------------
// RasterizedMapSensor, "entity" layer (sensor/map.py:120-192), SURVEY 8f N2: for the ego of every scenario an
// nh x nw occupancy grid in the ego's frame (rotated by heading + pi/2): cell = 1 iff the grid point lies strictly inside
// the bounding box of a present entity (the ego included).  One workgroup per scenario: the boxes' corners (fp64, the
// oracle's operation sequence) are staged in LDS once, then the threads stride over the grid points; the output
// [R][nh][nw] bytes is written coalesced.  np.linspace / numpy matmul arithmetic as probed (see the oracle).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double sg_linspace_at(double start, double stop, int n, int j)
{
    if (n > 1 && j == n - 1) return stop;
    const double step = n > 1 ? (stop - start) / (double)(n - 1) : 0.0;
    return (double)j * step + start;
}

#ifdef SG_UNIT_MAIN // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ __launch_bounds__(512) void raster_kernel(Params p, double width, double height, int nw, int nh,
                                                     unsigned char *out /*[R][nh][nw] at stride bytes per scenario*/,
                                                     int64_t stride)
{
    __shared__ double cor[8][512]; // (one thread per entity slot: 256 threads, 512 for scenarios of 257..512 entities)
    __shared__ unsigned char pres[512];
    __shared__ double ego_pose[4]; // x, y, sin(theta), cos(theta)
    __shared__ int near_n;
    const int r = blockIdx.x, e = threadIdx.x;
    const ScenStatic &ss = p.sstat[r];
    const uint32_t idx = (uint32_t)r * p.EP + (e < p.EP ? e : 0);
    const LanePtr st(p.stat + (size_t)(idx >> 6) * (ST_COUNT * 64), (idx & 63) * 8u);
    const LanePtr dy(p.dyn + (size_t)(idx >> 6) * ((size_t)p.FROWS * 64), (idx & 63) * 8u);
    const bool present = e < p.E && fld<uint64_t>(dy, SG_F_PRESENT) != 0;
    pres[e] = present;
    if (e == 0) near_n = 0;
    double C[8];
    if (present) {
        const double x = fld(dy, SG_F_POSE + 0), y = fld(dy, SG_F_POSE + 1), h = fld(dy, SG_F_POSE + 3);
        double s, c;
        sg_sincos(h, s, c);
        sg_corners(x, y, s, c, fld(st, ST_BW), fld(st, ST_BL), fld(st, ST_BCX), fld(st, ST_BCY), C);
    }
    if (e == ss.ego) {
        double s, c;
        sg_sincos(fld(dy, SG_F_POSE + 3) + 3.14159265358979311600e+00 / 2, s, c); // pose[3] + math.pi / 2
        ego_pose[0] = fld(dy, SG_F_POSE + 0); ego_pose[1] = fld(dy, SG_F_POSE + 1);
        ego_pose[2] = s; ego_pose[3] = c;
    }
    __syncthreads();
    const double ex = ego_pose[0], ey = ego_pose[1], s = ego_pose[2], c = ego_pose[3];
    if (present) {
        // only boxes that can reach the grid are tested per cell: every grid point lies within `reach` of the ego (the
        // grid's half diagonal, generously rounded up), every point of a box within the largest corner distance of its
        // first corner
        const double reach = 0.5 * (__builtin_fabs(width) + __builtin_fabs(height)) * 1.0000001 + 1e-6;
        double far = 0.0;
#pragma unroll
        for (int k = 1; k < 4; ++k) far = __builtin_fmax(far, __builtin_fabs(C[2 * k] - C[0]) + __builtin_fabs(C[2 * k + 1] - C[1]));
        const double dx = C[0] - ex, dyy = C[1] - ey, lim = reach + far * 1.0000001 + 1e-6 * (1.0 + __builtin_fabs(ex) + __builtin_fabs(ey));
        if (!(dx * dx + dyy * dyy > lim * lim)) { // NaN-safe: keeps the box
            const int q = atomicAdd(&near_n, 1);
#pragma unroll
            for (int k = 0; k < 8; ++k) cor[k][q] = C[k];
        }
    }
    __syncthreads();
    const bool ego_present = pres[ss.ego] != 0;
    const int nn = near_n;
    unsigned char *o = out + (size_t)r * stride;
    for (int q = e; q < nw * nh; q += (int)blockDim.x) {
        const int i = q / nw, j = q - i * nw;
        const double x0 = sg_linspace_at(-width / 2, width / 2, nw, j), x1 = sg_linspace_at(-height / 2, height / 2, nh, i);
        const double px = __builtin_fma(x1, -s, x0 * c) + ex, py = __builtin_fma(x1, c, x0 * s) + ey;
        bool hit = false;
        for (int k = 0; k < nn && !hit; ++k) {
            const double ax = cor[0][k], ay = cor[1][k], bx = cor[2][k], by = cor[3][k];
            const double cx = cor[4][k], cy = cor[5][k], dx = cor[6][k], dyy = cor[7][k];
            const double orient = (cx - ax) * (dyy - by) - (cy - ay) * (dx - bx);
            const double c0 = (bx - ax) * (py - ay) - (by - ay) * (px - ax);
            const double c1 = (cx - bx) * (py - by) - (cy - by) * (px - bx);
            const double c2 = (dx - cx) * (py - cy) - (dyy - cy) * (px - cx);
            const double c3 = (ax - dx) * (py - dyy) - (ay - dyy) * (px - dx);
            hit = orient > 0 ? (c0 > 0 && c1 > 0 && c2 > 0 && c3 > 0)
                             : (orient < 0 && c0 < 0 && c1 < 0 && c2 < 0 && c3 < 0);
        }
        o[q] = ego_present ? (unsigned char)hit : 0; // the reference sensor needs state.poses[entity]
    }
}
#endif // SG_UNIT_MAIN

// The road-surface layers of RasterizedMapSensor (sensor/map.py:194-271) on the same grid: one thread per grid point
// looks its cell up once for all requested layers; out[r][k] for the layers[k] != 0 (the entity layer is raster_kernel's).
#ifdef SG_UNIT_MAIN // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ __launch_bounds__(256) void raster_surface_kernel(Params p, RoadIndex R, double width, double height, int nw, int nh,
                                                             int n_layers, const int32_t *layers,
                                                             unsigned char *out /*[R][n_layers][nh][nw]*/)
{
    __shared__ double ego_pose[4];
    __shared__ int ego_present;
    const int r = blockIdx.x;
    const ScenStatic &ss = p.sstat[r];
    if (threadIdx.x == 0) {
        const uint32_t idx = (uint32_t)r * p.EP + ss.ego;
        const LanePtr dy(p.dyn + (size_t)(idx >> 6) * ((size_t)p.FROWS * 64), (idx & 63) * 8u);
        double s, c;
        sg_sincos(fld(dy, SG_F_POSE + 3) + 3.14159265358979311600e+00 / 2, s, c);
        ego_pose[0] = fld(dy, SG_F_POSE + 0); ego_pose[1] = fld(dy, SG_F_POSE + 1);
        ego_pose[2] = s; ego_pose[3] = c;
        ego_present = fld<uint64_t>(dy, SG_F_PRESENT) != 0;
    }
    __syncthreads();
    const double ex = ego_pose[0], ey = ego_pose[1], s = ego_pose[2], c = ego_pose[3];
    uint32_t want = 0;
    for (int k = 0; k < n_layers; ++k) want |= (uint32_t)layers[k];
    const int net = R.net_of_scen ? R.net_of_scen[r] : -1;
    unsigned char *o = out + (size_t)r * n_layers * nw * nh;
    for (int q = threadIdx.x; q < nw * nh; q += 256) {
        const int i = q / nw, j = q - i * nw;
        const double x0 = sg_linspace_at(-width / 2, width / 2, nw, j), x1 = sg_linspace_at(-height / 2, height / 2, nh, i);
        const double px = __builtin_fma(x1, -s, x0 * c) + ex, py = __builtin_fma(x1, c, x0 * s) + ey;
        const uint32_t in = ego_present ? rn_layers_at(R, net, want, px, py) : 0u;
        for (int k = 0; k < n_layers; ++k)
            if (layers[k]) o[(size_t)k * nw * nh + q] = (in & (uint32_t)layers[k]) != 0;
    }
}
#endif // SG_UNIT_MAIN

// ------------------------------------------------------------------------------------------------
// CollisionMetric.record_collision / get_collision_point / angle_between (metrics/collision.py:13-22, 81-203) for the
// events of Vehicle hazards, run when the events are read.  The reference's `.pose` attributes do not exist; the poses of
// the state at the event stand in: the ego's is stored with the event, the hazard's is its trajectory at the event time
// (replay entities: the same table segment and arithmetic as the rollout kernel).  Same operation sequence as the oracle:
// Sutherland-Hodgman clip of the ego box by the hazard box, area centroids over the triangle fan from the first vertex.
// ------------------------------------------------------------------------------------------------
__device__ inline double sg_pymod(double x, double m)
{
    double r = fmod(x, m);
    if (r != 0.0 && ((r < 0.0) != (m < 0.0))) r += m;
    return r;
}

__device__ inline bool sg_angle_between(double x, double lo, double hi)
{
    const double tau = 3.14159265358979311600e+00 * 2;
    x = sg_pymod(x, tau); lo = sg_pymod(lo, tau); hi = sg_pymod(hi, tau);
    return lo >= hi ? (lo < x || x <= hi) : (lo <= x && x < hi);
}

__device__ inline void sg_poly_centroid(const double *P, int n, double &cx, double &cy)
{
    double a2 = 0.0, sx = 0.0, sy = 0.0;
    for (int i = 1; i + 1 < n; ++i) {
        const double t2 = (P[2 * i] - P[0]) * (P[2 * i + 3] - P[1]) - (P[2 * i + 2] - P[0]) * (P[2 * i + 1] - P[1]);
        sx += t2 * (P[0] + P[2 * i] + P[2 * i + 2]);
        sy += t2 * (P[1] + P[2 * i + 1] + P[2 * i + 3]);
        a2 += t2;
    }
    if (a2 != 0.0) { cx = sx / 3 / a2; cy = sy / 3 / a2; return; }
    sx = sy = 0.0;
    for (int i = 0; i < n; ++i) { sx += P[2 * i]; sy += P[2 * i + 1]; }
    cx = n ? sx / n : __builtin_nan("");
    cy = n ? sy / n : __builtin_nan("");
}

__device__ inline int sg_clip_quads(const double *S, const double *C, double *A /*[16]*/)
{
    double B[16];
    int na = 4;
    for (int i = 0; i < 8; ++i) A[i] = S[i];
    double orient = 0.0;
    for (int k = 0; k < 4; ++k) { const int m = (k + 1) & 3; orient += C[2 * k] * C[2 * m + 1] - C[2 * m] * C[2 * k + 1]; }
    const double sgn = orient >= 0 ? 1.0 : -1.0;
    for (int k = 0; k < 4 && na > 0; ++k) {
        const int m = (k + 1) & 3;
        const double ex = C[2 * m] - C[2 * k], ey = C[2 * m + 1] - C[2 * k + 1];
        int nb = 0;
        for (int i = 0; i < na; ++i) {
            const int j = (i + 1) % na;
            const double di = sgn * (ex * (A[2 * i + 1] - C[2 * k + 1]) - ey * (A[2 * i] - C[2 * k]));
            const double dj = sgn * (ex * (A[2 * j + 1] - C[2 * k + 1]) - ey * (A[2 * j] - C[2 * k]));
            if (di >= 0) { B[2 * nb] = A[2 * i]; B[2 * nb + 1] = A[2 * i + 1]; ++nb; }
            if ((di > 0 && dj < 0) || (di < 0 && dj > 0)) {
                const double u = di / (di - dj);
                B[2 * nb] = A[2 * i] + u * (A[2 * j] - A[2 * i]);
                B[2 * nb + 1] = A[2 * i + 1] + u * (A[2 * j + 1] - A[2 * i + 1]);
                ++nb;
            }
        }
        for (int i = 0; i < 2 * nb; ++i) A[i] = B[i];
        na = nb;
    }
    return na;
}

// CollisionPoints: 0 front, 1 front_corner, 2 side, 3 back, 4 back_corner
__device__ inline int sg_collision_point_class(const double *box8, double angle, double heading, double c_tol)
{
    double bx, by, cor[4];
    sg_poly_centroid(box8, 4, bx, by);
    for (int k = 0; k < 4; ++k) cor[k] = sg_atan2(box8[2 * k + 1] - by, box8[2 * k] - bx) - heading;
    if (sg_angle_between(angle, cor[1] - c_tol, cor[1] + c_tol) || sg_angle_between(angle, cor[2] - c_tol, cor[2] + c_tol)) return 1;
    if (sg_angle_between(angle, cor[0] - c_tol, cor[0] + c_tol) || sg_angle_between(angle, cor[3] - c_tol, cor[3] + c_tol)) return 4;
    if (sg_angle_between(angle, cor[0] + c_tol, cor[3] - c_tol)) return 3;
    if (sg_angle_between(angle, cor[2] - c_tol, cor[1] + c_tol)) return 0;
    return 2;
}

__device__ inline int sg_classify_collision(const double *eb, double ex, double ey, double eh, const double *hb, double hx,
                                            double hy, double hh, double c_tol, double &px, double &py, double &collision_angle)
{
    const double pi = 3.14159265358979311600e+00, tau = pi * 2;
    double clip[16];
    const int n = sg_clip_quads(eb, hb, clip);
    sg_poly_centroid(clip, n, px, py); // CollisionPointMetric.record_collision_position, metrics/collision.py:242-253
    collision_angle = sg_pymod(hh - eh, tau);
    const double ego_angle = sg_pymod(sg_atan2(py - ey, px - ex) - eh, tau);
    const double haz_angle = sg_pymod(sg_atan2(py - hy, px - hx) - hh, tau);
    const int ep = sg_collision_point_class(eb, ego_angle, eh, c_tol), hp = sg_collision_point_class(hb, haz_angle, hh, c_tol);
    const bool ef = ep == 0 || ep == 1, ebk = ep == 3 || ep == 4, hf = hp == 0 || hp == 1, hbk = hp == 3 || hp == 4;
    const bool cross = sg_angle_between(collision_angle, pi / 4, 3 * pi / 4) || sg_angle_between(collision_angle, 5 * pi / 4, 7 * pi / 4);
    if (ef && hf) return cross ? 1 : (sg_angle_between(collision_angle, 7 * pi / 4, pi / 4) ? 4 : 2);
    if ((ef || ebk) && (hf || hbk)) return cross ? 1 : 3;
    if (ef || ebk || hf || hbk) return cross ? 1 : 4;
    return 4;
}

// Right after a table-variant launch, while its controller table is still there: the events it recorded for Vehicle hazards
// (type packed with k, the step inside the launch) take the controlled ego's pose at that step from the table row and become
// ordinary pending events.  A few loads and stores per event; the classification itself waits for sg_read_metrics.
// (`tg`: the block groups of that launch -- the scenario's group says which buffer its rows are in)
#ifdef SG_UNIT_MAIN // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ __launch_bounds__(64) void event_ego_pose_kernel(Params p, TabGroups tg)
{
    const int r = blockIdx.x;
    const int n = min(p.sdyn[r].n_events, p.ev_cap);
    if (n == 0) return;
    int n_launch;
    const double *tab;
    if (!tg.pick((unsigned)(((size_t)r * p.EP) >> 6), n_launch, tab)) return; // (its group sat the launch out: nothing packed)
    const uint32_t eidx = (uint32_t)r * p.EP + p.sstat[r].ego;
    const LanePtr est(p.stat + (size_t)(eidx >> 6) * (ST_COUNT * 64), (eidx & 63) * 8u);
    const int64_t ectl = fld<int64_t>(est, ST_CTL);
    const int ekind = (int)(fld<int64_t>(est, ST_META) & 0xff);
    const bool from_tab = ectl >= 0 && (ekind == SG_KIND_AGENT_PID || ekind == SG_KIND_AGENT_VEHICLE);
    for (int i = threadIdx.x; i < n; i += 64) {
        sg_event &ev = p.events[(size_t)r * p.ev_cap + i];
        if (ev.type < 16) continue; // not packed: recorded by another launch
        const int k_launch = (ev.type >> 4) - 1, base = ev.type & 15;
        if (from_tab) {
            const double *row = tab + ((size_t)ectl * (size_t)(p.tab_steps + 1) + (size_t)k_launch) * CT_W;
            double *ep = p.ev_pose + ((size_t)r * p.ev_cap + i) * 3;
            ep[0] = row[CT_X]; ep[1] = row[CT_Y]; ep[2] = row[CT_H];
        }
        {   // a hazard that is a controlled agent: its pose at that step is a row of the table as well
            const uint32_t hidx = (uint32_t)r * p.EP + ev.other;
            const LanePtr hst(p.stat + (size_t)(hidx >> 6) * (ST_COUNT * 64), (hidx & 63) * 8u);
            const int64_t hctl = fld<int64_t>(hst, ST_CTL);
            const int hkind = (int)(fld<int64_t>(hst, ST_META) & 0xff);
            double *hp = p.ev_hpose + ((size_t)r * p.ev_cap + i) * 3;
            if (hctl >= 0 && (hkind == SG_KIND_AGENT_PID || hkind == SG_KIND_AGENT_VEHICLE)) {
                const double *row = tab + ((size_t)hctl * (size_t)(p.tab_steps + 1) + (size_t)k_launch) * CT_W;
                hp[0] = row[CT_X]; hp[1] = row[CT_Y]; hp[2] = row[CT_H];
            } else {
                hp[0] = hp[1] = hp[2] = __builtin_nan("");
            }
        }
        ev.type = base == 15 ? -1 : base;
    }
}
#endif // SG_UNIT_MAIN

// one thread per (scenario, event slot): pending events (-1) get their type, or -2 when the hazard's pose cannot be
// re-evaluated.  Ego pose: its trajectory (replay agents), else the pose stored with the event.
#ifdef SG_UNIT_MAIN // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ __launch_bounds__(64) void classify_events_kernel(Params p, double c_tol)
{
    const int r = blockIdx.x;
    const ScenStatic &ss = p.sstat[r];
    const int n = min(p.sdyn[r].n_events, p.ev_cap);
    for (int i = threadIdx.x; i < n; i += 64) {
        sg_event &ev = p.events[(size_t)r * p.ev_cap + i];
        if (ev.reserved != 0) continue; // done by an earlier read
        if (ev.type >= 16) ev.type = (ev.type & 15) == 15 ? -1 : (ev.type & 15); // packed by a table-variant launch without controlled lanes
        const bool vehicle = ev.type == -1;
        ev.reserved = 1;
        double *pt = p.ev_pose + ((size_t)r * p.ev_cap + i) * 3; // in: ego pose of the event, out: collision point + angle
        const uint32_t hidx = (uint32_t)r * p.EP + ev.other, eidx = (uint32_t)r * p.EP + ss.ego;
        const LanePtr hst(p.stat + (size_t)(hidx >> 6) * (ST_COUNT * 64), (hidx & 63) * 8u);
        const LanePtr est(p.stat + (size_t)(eidx >> 6) * (ST_COUNT * 64), (eidx & 63) * 8u);
        const int64_t meta = fld<int64_t>(hst, ST_META);
        const int kind = (int)(meta & 0xff);
        double hp[6];
        if (kind == SG_KIND_REPLAY) { // BatchReplayEntity: the union-grid segment containing t, as the rollout kernel has it
            Table T = lane_table(p, kind, ss, ev.other, hst);
            Segment S;
            S.cur = seg_locate(T, ev.t);
            seg_load(T, S);
            const double dq = ev.t - S.x_lo;
            for (int c = 0; c < 6; ++c) hp[c] = S.sl[c] * dq + S.ylo[c];
        } else if (kind == SG_KIND_AGENT_REPLAY) {
            own_position_clamped(p.knots + fld<int64_t>(hst, ST_KNOT_OFF) * 7, (int)(meta >> 32), ev.t, hp);
        } else { // a controlled hazard: the pose it left beside the event (rollout kernel / event_ego_pose_kernel)
            const double *hq = p.ev_hpose + ((size_t)r * p.ev_cap + i) * 3;
            if (!(hq[0] == hq[0])) { // not saved (wide tiles with in-kernel controllers, aliased geometries)
                if (vehicle) ev.type = -2;
                pt[0] = pt[1] = pt[2] = __builtin_nan("");
                continue;
            }
            hp[0] = hq[0]; hp[1] = hq[1]; hp[3] = hq[2];
            hp[2] = hp[4] = hp[5] = 0.0;
        }
        const int64_t emeta = fld<int64_t>(est, ST_META);
        double ex = pt[0], ey = pt[1], eh = pt[2];
        if ((int)(emeta & 0xff) == SG_KIND_AGENT_REPLAY) {
            double q[6];
            own_position_clamped(p.knots + fld<int64_t>(est, ST_KNOT_OFF) * 7, (int)(emeta >> 32), ev.t, q);
            ex = q[0]; ey = q[1]; eh = q[3];
        }
        double s, c, EB[8], HB[8];
        sg_sincos(eh, s, c);
        sg_corners(ex, ey, s, c, fld(est, ST_BW), fld(est, ST_BL), fld(est, ST_BCX), fld(est, ST_BCY), EB);
        sg_sincos(hp[3], s, c);
        sg_corners(hp[0], hp[1], s, c, fld(hst, ST_BW), fld(hst, ST_BL), fld(hst, ST_BCX), fld(hst, ST_BCY), HB);
        double cpx, cpy, cang;
        const int cls = sg_classify_collision(EB, ex, ey, eh, HB, hp[0], hp[1], hp[3], c_tol, cpx, cpy, cang);
        if (vehicle) ev.type = cls;
        pt[0] = cpx; pt[1] = cpy; pt[2] = cang;
    }
}
#endif // SG_UNIT_MAIN

// rss_state [NE] = found | last << 8; code [NE]: 0 safe, 1 lateral, 2 longitudinal, 3 both, 4 unsafe_lateral,
// 5 unsafe_longitudinal, 6 found, -1 not updated; safe [NE][2] = lateral, longitudinal
// seen [R]: State.n_steps at the scenario's latest update -- a scenario that did not step since (it is done) is left alone,
// as the reference stops calling the callback once its rollout loop has ended
#ifdef SG_UNIT_MAIN // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ __launch_bounds__(512) void rss_kernel(Params p, int reset, int32_t *rss_state, int32_t *code, double *safe, int32_t *seen)
{
    __shared__ double ego[8]; // x, y, heading, vx, vy, width, length, present
    const int r = blockIdx.x, e = threadIdx.x;
    const ScenStatic &ss = p.sstat[r];
    const uint32_t idx = (uint32_t)r * p.EP + (e < p.EP ? e : 0);
    const LanePtr st(p.stat + (size_t)(idx >> 6) * (ST_COUNT * 64), (idx & 63) * 8u);
    const LanePtr dy(p.dyn + (size_t)(idx >> 6) * ((size_t)p.FROWS * 64), (idx & 63) * 8u);
    const bool in = e < p.E;
    const bool present = in && fld<uint64_t>(dy, SG_F_PRESENT) != 0;
    double hp[4] = {0, 0, 0, 0}, hv[2] = {0, 0};
    if (in) {
        hp[0] = fld(dy, SG_F_POSE + 0); hp[1] = fld(dy, SG_F_POSE + 1); hp[3] = fld(dy, SG_F_POSE + 3);
        hv[0] = fld(dy, SG_F_VEL + 0); hv[1] = fld(dy, SG_F_VEL + 1);
    }
    if (e == ss.ego) {
        ego[0] = hp[0]; ego[1] = hp[1]; ego[2] = hp[3]; ego[3] = hv[0]; ego[4] = hv[1];
        ego[5] = fld(st, ST_BW); ego[6] = fld(st, ST_BL); ego[7] = present ? 1.0 : 0.0;
    }
    const int steps_now = p.sdyn[r].n_steps;
    const bool stale = !reset && seen[r] == steps_now;
    __syncthreads();
    if (e == 0) seen[r] = steps_now;
    if (!in || stale) return;
    int32_t state = reset ? 0 : rss_state[idx];
    int cd = -1;
    double s_lat = __builtin_nan(""), s_long = __builtin_nan("");
    const bool skip = p.sdyn[r].t == 0.0 || ego[7] == 0.0 || e == ss.ego || !present; // callback.py:76-78
    if (!skip)
        rss_entity(ego[0], ego[1], ego[2], ego[3], ego[4], ego[5], ego[6], hp[0], hp[1], hp[3], hv[0], hv[1], fld(st, ST_BW),
                   fld(st, ST_BL), fld(st, ST_BCX), fld(st, ST_BCY), state, cd, s_lat, s_long);
    rss_state[idx] = state;
    code[idx] = cd;
    safe[(size_t)idx * 2] = s_lat;
    safe[(size_t)idx * 2 + 1] = s_long;
}
#endif // SG_UNIT_MAIN

// The queued line tests of one rollout_kernel_rss launch (see RssQueue): block w = the queue of rollout wavefront w, whose
// lane l carries entity index w * 64 + l.
// (tg: the blocks of that launch -- one pipeline's part of the batch, launch_rollout; else all of them)
#ifdef SG_UNIT_RSS_LINES // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ __launch_bounds__(64) void rss_lines_kernel(Params p, TabGroups tg)
{
    __shared__ RssQueue q;
    const RssQueueLds ql = (RssQueueLds)&q;
    const int lane = threadIdx.x;
    const size_t w = tg.map(blockIdx.x);
    const int n = p.rssq_n[w];
    const uint32_t idx = (uint32_t)(w * 64 + lane);
    int32_t st = p.rss_state[idx];
    if (n == 0 && !sg_any(st & RSS_ST_PENDING)) return;
    ql->lastword[lane] = 0;
    ql->stepcd[lane] = 0;
    ql->hits[lane] = 0;
    const double *rec0 = p.rssq + w * (size_t)p.rssq_cap * RSSQ_REC;
    for (int g0 = 0; g0 < n; g0 += RSSQ_CAP) {
        const int m = min(RSSQ_CAP, n - g0);
        if (lane < m) {
            const double2 *rec = reinterpret_cast<const double2 *>(rec0 + (size_t)(g0 + lane) * RSSQ_REC);
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const double2 v = rec[k];
                ql->q[2 * k][lane] = v.x;
                ql->q[2 * k + 1][lane] = v.y;
            }
            const uint64_t mk = (uint64_t)__double_as_longlong(rec[5].x);
            ql->meta[lane] = (int)(uint32_t)mk;
            ql->key[lane] = (unsigned)(mk >> 32);
        }
        tile_sync<1>();
        rss_flush_body(ql, m);
    }
    tile_sync<1>();
    const unsigned lw = ql->lastword[lane], sc = ql->stepcd[lane];
    int cd = p.rss_code[idx];
    const int32_t st0 = st;
    const int cd0 = cd;
    if (lw) st = (st & ~0xff00) | (int)(lw & 3) << 8;
    if (st & RSS_ST_PENDING) { // the entity entered the buffer during the launch: unsafe_distance, callback.py:196-213
        const int last = (st >> 8) & 0xff;
        const int cls = last == 1 ? 5 : (last == 2 ? 4 : ((st & RSS_ST_AB) ? 5 : 4));
        st = (st & 0xff00) | (cls == 4 ? 1 : 2);
        if (cd == RSS_CD_ISECT) cd = cls;
    }
    if (cd <= -4) cd = (sc >> 3) == (unsigned)(-4 - cd) ? (int)(sc & 7) : 0; // the latest update's line tests were queued
    if (st != st0) p.rss_state[idx] = st;
    if (cd != cd0) p.rss_code[idx] = cd;
}
#endif // SG_UNIT_RSS_LINES

// The observation of one RL tick in ONE launch (sg_tick): every requested map layer -- the entity layer of raster_kernel and
// the surface layers of raster_surface_kernel, same arithmetic, the grid point computed once -- and the terminal flags of
// terminal_flags_kernel.  One workgroup per scenario.  has_road: road networks are set (else the surface layers are empty).
#ifdef SG_UNIT_MAIN // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ __launch_bounds__(512) void observe_kernel(Params p, RoadIndex R, int has_road, double width, double height, int nw,
                                                      int nh, int n_layers, const int32_t *layers,
                                                      unsigned char *out /*[R][n_layers][nh][nw]*/, uint32_t *flags /*[R]*/)
{
    __shared__ double cor[8][512]; // (one thread per entity slot: 256 threads, 512 for scenarios of 257..512 entities)
    __shared__ double ego_pose[4]; // x, y, sin(theta), cos(theta)
    __shared__ int near_n, ego_present, any_coll;
    const int r = blockIdx.x, e = threadIdx.x;
    const ScenStatic &ss = p.sstat[r];
    const uint32_t idx = (uint32_t)r * p.EP + (e < p.EP ? e : 0);
    const LanePtr st(p.stat + (size_t)(idx >> 6) * (ST_COUNT * 64), (idx & 63) * 8u);
    const LanePtr dy(p.dyn + (size_t)(idx >> 6) * ((size_t)p.FROWS * 64), (idx & 63) * 8u);
    const bool present = e < p.E && fld<uint64_t>(dy, SG_F_PRESENT) != 0;
    if (e == 0) { near_n = 0; ego_present = 0; any_coll = 0; }
    __syncthreads();
    double C[8], x = 0.0, y = 0.0;
    bool mine = false;
    if (e < p.E) {
        const int W = p.FROWS - SG_F_COLL;
        for (int w = 0; w < W; ++w) mine = mine || fld<uint64_t>(dy, SG_F_COLL + w) != 0;
        if (present && mine) any_coll = 1;
    }
    if (present) {
        x = fld(dy, SG_F_POSE + 0); y = fld(dy, SG_F_POSE + 1);
        const double h = fld(dy, SG_F_POSE + 3);
        double s, c;
        sg_sincos(h, s, c);
        sg_corners(x, y, s, c, fld(st, ST_BW), fld(st, ST_BL), fld(st, ST_BCX), fld(st, ST_BCY), C);
    }
    if (e == ss.ego) {
        double s, c;
        sg_sincos(fld(dy, SG_F_POSE + 3) + 3.14159265358979311600e+00 / 2, s, c); // pose[3] + math.pi / 2
        ego_pose[0] = fld(dy, SG_F_POSE + 0); ego_pose[1] = fld(dy, SG_F_POSE + 1);
        ego_pose[2] = s; ego_pose[3] = c;
        ego_present = present;
    }
    const int net = (has_road && R.net_of_scen) ? R.net_of_scen[r] : -1;
    if (e == 0 && flags) { // TERMINAL_CONDITIONS of entities[0], state/state.py:397-408 (terminal_flags_kernel)
        const sg_scenario_state &sd = p.sdyn[r];
        uint32_t bits = 0;
        if (sd.t + (sd.t - sd.prev_t) > ss.length) bits |= SG_TERM_MAX_LENGTH;
        if (present && mine) bits |= SG_TERM_EGO_COLLISION;
        bool on_road = false;
        if (present && has_road) on_road = (rn_layers_at(R, net, SG_LAYER_DRIVEABLE, x, y) & SG_LAYER_DRIVEABLE) != 0;
        if (!on_road) bits |= SG_TERM_EGO_OFF_ROAD;
        flags[r] = bits; // SG_TERM_COLLISION joins below, once every entity has reported
    }
    __syncthreads();
    if (e == 0 && flags && any_coll) flags[r] |= SG_TERM_COLLISION;
    const double ex = ego_pose[0], ey = ego_pose[1], s = ego_pose[2], c = ego_pose[3];
    bool want_entity = false;
    uint32_t want = 0;
    for (int k = 0; k < n_layers; ++k) { want_entity = want_entity || layers[k] == 0; want |= (uint32_t)layers[k]; }
    if (present && want_entity) { // the boxes that can reach the grid (raster_kernel)
        const double reach = 0.5 * (__builtin_fabs(width) + __builtin_fabs(height)) * 1.0000001 + 1e-6;
        double far = 0.0;
#pragma unroll
        for (int k = 1; k < 4; ++k) far = __builtin_fmax(far, __builtin_fabs(C[2 * k] - C[0]) + __builtin_fabs(C[2 * k + 1] - C[1]));
        const double dx = C[0] - ex, dyy = C[1] - ey, lim = reach + far * 1.0000001 + 1e-6 * (1.0 + __builtin_fabs(ex) + __builtin_fabs(ey));
        if (!(dx * dx + dyy * dyy > lim * lim)) {
            const int q = atomicAdd(&near_n, 1);
#pragma unroll
            for (int k = 0; k < 8; ++k) cor[k][q] = C[k];
        }
    }
    __syncthreads();
    const int nn = near_n;
    const bool ego_pres = ego_present != 0;
    unsigned char *o = out + (size_t)r * n_layers * nw * nh;
    for (int q = e; q < nw * nh; q += (int)blockDim.x) {
        const int i = q / nw, j = q - i * nw;
        const double x0 = sg_linspace_at(-width / 2, width / 2, nw, j), x1 = sg_linspace_at(-height / 2, height / 2, nh, i);
        const double px = __builtin_fma(x1, -s, x0 * c) + ex, py = __builtin_fma(x1, c, x0 * s) + ey;
        bool hit = false;
        for (int k = 0; k < nn && !hit; ++k) {
            const double ax = cor[0][k], ay = cor[1][k], bx = cor[2][k], by = cor[3][k];
            const double cx = cor[4][k], cy = cor[5][k], dx = cor[6][k], dyy = cor[7][k];
            const double orient = (cx - ax) * (dyy - by) - (cy - ay) * (dx - bx);
            const double c0 = (bx - ax) * (py - ay) - (by - ay) * (px - ax);
            const double c1 = (cx - bx) * (py - by) - (cy - by) * (px - bx);
            const double c2 = (dx - cx) * (py - cy) - (dyy - cy) * (px - cx);
            const double c3 = (ax - dx) * (py - dyy) - (ay - dyy) * (px - dx);
            hit = orient > 0 ? (c0 > 0 && c1 > 0 && c2 > 0 && c3 > 0)
                             : (orient < 0 && c0 < 0 && c1 < 0 && c2 < 0 && c3 < 0);
        }
        const uint32_t in = (ego_pres && want && has_road) ? rn_layers_at(R, net, want, px, py) : 0u;
        for (int k = 0; k < n_layers; ++k)
            o[(size_t)k * nw * nh + q] = layers[k] == 0 ? (unsigned char)(ego_pres && hit) : (unsigned char)((in & (uint32_t)layers[k]) != 0);
    }
}
#endif // SG_UNIT_MAIN

// TERMINAL_CONDITIONS (state/state.py:397-408), all four evaluated on the CURRENT state of every scenario, whatever the
// handle's terminal mask says: out[r] = SG_TERM_* bits.  The reward of the reference's RL agent asks exactly this of a
// done state (integrations/openaigym.py:300-310).  One wavefront per scenario.
#ifdef SG_UNIT_MAIN // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ __launch_bounds__(64) void terminal_flags_kernel(Params p, double timestep, uint32_t *out)
{
    const int r = blockIdx.x, lane = threadIdx.x;
    const sg_scenario_state &sd = p.sdyn[r];
    const int W = p.FROWS - SG_F_COLL;
    bool any_coll = false, ego_coll = false, e0_present = false;
    double x0 = 0.0, y0 = 0.0;
    for (int e = lane; e < p.E; e += 64) {
        const uint32_t idx = (uint32_t)r * p.EP + e;
        const LanePtr dy(p.dyn + (size_t)(idx >> 6) * ((size_t)p.FROWS * 64), (idx & 63) * 8u);
        const bool present = fld<uint64_t>(dy, SG_F_PRESENT) != 0;
        bool mine = false;
        for (int w = 0; w < W; ++w) mine = mine || fld<uint64_t>(dy, SG_F_COLL + w) != 0;
        any_coll = any_coll || (present && mine);
        if (e == 0) {
            e0_present = present;
            ego_coll = present && mine;
            x0 = fld(dy, SG_F_POSE + 0);
            y0 = fld(dy, SG_F_POSE + 1);
        }
    }
    uint32_t bits = 0;
    if (sd.t + (sd.t - sd.prev_t) > p.sstat[r].length) bits |= SG_TERM_MAX_LENGTH; // s.t + s.dt > length, State.dt = t - prev_t
    if (sg_any(any_coll)) bits |= SG_TERM_COLLISION;
    if (lane == 0) {
        if (ego_coll) bits |= SG_TERM_EGO_COLLISION;
        bool on_road = false;
        if (e0_present && p.road) {
            const RoadIndex RI = *p.road;
            on_road = (rn_layers_at(RI, RI.net_of_scen[r], SG_LAYER_DRIVEABLE, x0, y0) & SG_LAYER_DRIVEABLE) != 0;
        }
        if (!on_road) bits |= SG_TERM_EGO_OFF_ROAD;
        out[r] = bits;
    }
}
#endif // SG_UNIT_MAIN

// sg_debug_trig32: the broad phase's hardware sin/cos, exposed so that the parity tests can bound its error
#ifdef SG_UNIT_MAIN // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ void trig32_kernel(const double *h, float *s, float *c, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sg_sincos_f32(h[i], s[i], c[i]);
}
#endif // SG_UNIT_MAIN

} // namespace sg
