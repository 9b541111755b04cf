// sgym_core.hpp -- Structures, lane pointers, fp64 / fp32 math shared with the oracle by restatement, RecipDiv, knot tables, box geometry, the LDS tile.
// Part of the gfx950 device code of the batched rollout engine; included by sgym_device.hpp (in order: every part builds on
// the ones before it), never on its own.
#pragma once

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
    const double *imp_aux;          // [n][4] per such edge: bx - ax, by - ay, ~1 / |b - a|^2, 0 (the filter of ped_boundary_terms)
    const double *imp_m;            // [n_nets] the largest |coordinate| of the network's impenetrable edges (its error margin)
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
    int ped_behaviour;       // sg_set_ped_behaviour: 0 SocialForce, 1 RandomWalk (never with the crowd variants)
    // per-agent behaviour models (sg_set_ped_models; n_ped_models <= 1: `sf` / `ped_behaviour` / `noise_std_*` above are the model)
    int n_ped_models;
    const double *ped_models; // [n_ped_models][PM_W]: behaviour, the 12 doubles of sg_social_force, std_lon, std_lat
    const int32_t *model_of;  // [NE] model of every padded entity slot (0 where there is none)
#ifdef SG_PHASE_TIMERS
    unsigned long long *phase_cycles; // [16] experiment builds: s_memtime cycles per phase of the step, summed over wavefronts
#endif
};

enum { PM_BEHAVIOUR = 0, PM_SF = 1, PM_STD_LON = 13, PM_STD_LAT = 14, PM_W = 16 }; // doubles of one row of Params::ped_models
static_assert(sizeof(sg_social_force) == 12 * sizeof(double), "sg_social_force is twelve doubles");

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
static __constant__ double SG_TRIG[48] = {
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
    // 32-42: atan polynomial A0..A10 (sg_atan_pos)
    3.33333333333329318027e-01, -1.99999999998764832476e-01, 1.42857142725034663711e-01, -1.11111104054623557880e-01,
    9.09088713343650656196e-02, -7.69187620504482999495e-02, 6.66107313738753120669e-02, -5.83357013379057348645e-02,
    4.97687799461593236017e-02, -3.65315727442169155270e-02, 1.62858201153657823623e-02, 0.0, 0.0, 0.0, 0.0, 0.0,
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
template <int NS, bool PED, bool CROWD = false, bool ROADTAB = false>
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
    // (crowd variants: one scenario per workgroup) the ego lane's accumulators -- EgoAvgSpeed, EgoMaxSpeed, EgoAvgSpeed.t; CollisionMetric.last_timestep; the
    // event count -- wait here between their two uses per step instead of in 15 registers of every lane
    double ego_m[CROWD ? 3 : 1];
    unsigned long long ego_last[CROWD ? (NS + 63) / 64 : 1];
    int ego_nev[2];
    // ROADTAB (rollout_kernel_crowd / _models: one scenario per workgroup): the ring edges of the scenario's buildings for the
    // boundary terms of the social force, staged once per launch (rollout_body_l; ped_boundary_terms): [k][5] = ax, ay, bx, by,
    // ~1 / |b - a|^2; road_m = the network's largest |coordinate|; road_info = edges staged (-1: none -- no network, no
    // buildings, more than 64 edges: device memory), the network, its flags.  Not in the riders variant: two of its
    // workgroups and one of control_kernel_riders (7 KB) share a compute unit's 160 KB, and these 2.5 KB would evict the latter
    // (measured: c5mix 781 -> 902 ms).
    // (crowd variants of one and four wavefronts; the two-wavefront tile has no kilobyte to spare: four workgroups per unit) the
    // vertex table of Point.buffer(r) for the radius rule's thin ring (sg_in_radius): from device memory its four vertices
    // were four memory latencies in a row, met by about one pair-loop round in seven
    static constexpr bool GON = CROWD && NS != 128;
    double gon[GON ? 128 : 1];
    static constexpr bool ROAD_TAB = ROADTAB;
    static constexpr int ROAD_EDGES = 64;
    double road_tab[ROADTAB ? ROAD_EDGES * 5 : 1];
    double road_m;
    RoadNet road_net;   // the network's grid header (rn_probe)
    int road_info[4];

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

} // namespace sg
