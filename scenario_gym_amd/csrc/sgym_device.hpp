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
#include <stdint.h>

#include "../../include/sgym.h"

#ifndef SG_WAVES_PER_SIMD
#define SG_WAVES_PER_SIMD 2 // register budget of the rollout kernel: 512 / SG_WAVES_PER_SIMD VGPRs per lane
#endif

namespace sg {

struct Params {
    // geometry of the batch
    int R, E, EP, W;
    int persist;
    unsigned term_mask;
    int rec_cap, ev_cap;
    // static scenario data (device pointers)
    const int32_t *kind, *etype;
    const double *bbox[4];
    const double *min_t, *max_t;
    const int64_t *knot_off;
    const int32_t *knot_n;
    const double *knots;
    const double *ctrl[9];
    const int32_t *ego;
    const double *t0, *length;
    const int64_t *grid_off;
    const int32_t *grid_n;
    const double *grid_t;
    double *grid_y;
    // mutable state
    double *pose[6], *vel[6], *dist;
    uint64_t *coll;
    uint8_t *present;
    double *cs[4];
    double *t, *prev_t;
    int32_t *done, *n_steps;
    double *m_avg, *m_max, *m_t, *m_dist;
    uint64_t *last_row;
    int32_t *n_events;
    sg_event *events;
    double *rec_t, *rec_pose;
    int32_t *rec_rows;
};

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

__device__ __forceinline__ void sg_sincos(double x, double &s, double &c)
{
    if (!(__builtin_fabs(x) < 1.0e5)) {
        double2 sc = sg_sincos_slow(x);
        s = sc.x;
        c = sc.y;
        return;
    }
    const double INV_PIO2 = 6.36619772367581382433e-01, PIO2_1 = 1.57079632673412561417e+00,
                 PIO2_2 = 6.07710050630396597660e-11, PIO2_2T = 2.02226624879595063154e-21;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double fn = __builtin_rint(x * INV_PIO2);
    int n = (int)fn;
    double t = x - fn * PIO2_1;
    double w = fn * PIO2_2;
    double r = t - w;
    w = fn * PIO2_2T - ((t - r) - w);
    double y0 = r - w;
    double y1 = (r - y0) - w;
    double z = y0 * y0;
    double v = z * y0;
    double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    double ks = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * S1);
    double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
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

__device__ __forceinline__ double sg_pred(double x) // nextafter(x, -inf) for finite x
{
    long long b = __double_as_longlong(x);
    if (x > 0.0) return __longlong_as_double(b - 1);
    if (x < 0.0) return __longlong_as_double(b + 1);
    return -4.9406564584124654e-324;
}

__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src, 64); }

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
        double x_lo = kn[0], x_hi = kn[0] + 1e-3;
        for (int c = 0; c < 6; ++c) {
            double slope = (kn[1 + c] - kn[1 + c]) / (x_hi - x_lo);
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

// ------------------------------------------------------------------------------------------------
// controllers
// ------------------------------------------------------------------------------------------------
struct CtrlState { double speed, e_lon_prev, e_lat_prev, e_lon_int; };

// VehicleController._step (controller.py:105-140); sin_h/cos_h of the current heading come from
// the previous step's corner computation.
__device__ __forceinline__ void vehicle_step(CtrlState &cs, const Params &p, size_t idx, double l,
                                             double dt, double accel, double steer, double sin_h,
                                             double cos_h, double *pose)
{
    double max_steer = p.ctrl[SG_C_MAX_STEER][idx], max_accel = p.ctrl[SG_C_MAX_ACCEL][idx];
    double max_speed = p.ctrl[SG_C_MAX_SPEED][idx], allow_rev = p.ctrl[SG_C_ALLOW_REVERSE][idx];
    accel = __builtin_fmin(__builtin_fmax(accel, -max_accel), max_accel);
    steer = __builtin_fmin(__builtin_fmax(steer, -max_steer), max_steer);
    double ss, sc;
    sg_sincos(steer, ss, sc);
    double dx = cs.speed * cos_h;
    double dy = cs.speed * sin_h;
    double dh = cs.speed * (ss / sc) / l;
    pose[0] += dx * dt;
    pose[1] += dy * dt;
    pose[3] += dh * dt;
    double speed = cs.speed + accel * dt;
    if (allow_rev == 0.0) speed = __builtin_fmax(0.0, speed);
    if (max_speed == max_speed) speed = __builtin_fmin(max_speed, speed);
    cs.speed = speed;
}

// PIDController._step (controller.py:205-258)
__device__ __forceinline__ void pid_step(CtrlState &cs, const Params &p, size_t idx, double l,
                                         double state_dt, double dt, double tx, double ty,
                                         double sin_h, double cos_h, double *pose)
{
    double e0 = tx - pose[0], e1 = ty - pose[1];
    double e_lon = cos_h * e0 + sin_h * e1;
    double e_lat = -sin_h * e0 + cos_h * e1;
    double speed = cs.speed, gain;
    if (speed > 5.0 && speed <= 15) gain = 1.0 - 0.9 * (speed - 5.0) / 10.0;
    else if (speed > 15) gain = 0.1;
    else gain = 1.0;
    double e_lat_D = (e_lat - cs.e_lat_prev) / state_dt;
    double kp = p.ctrl[SG_C_STEER_KP][idx] * gain, kd = p.ctrl[SG_C_STEER_KD][idx] * gain;
    double steer = kp * e_lat + kd * e_lat_D;
    double e_lon_D = (e_lon - cs.e_lon_prev) / state_dt;
    double e_lon_I = cs.e_lon_int + e_lon * state_dt;
    double accel = 0.0;
    if (__builtin_fabs(e_lon) > 0.1)
        accel = p.ctrl[SG_C_ACCEL_KP][idx] * e_lon + p.ctrl[SG_C_ACCEL_KD][idx] * e_lon_D +
                p.ctrl[SG_C_ACCEL_KI][idx] * e_lon_I;
    cs.e_lat_prev = e_lat;
    cs.e_lon_prev = e_lon;
    cs.e_lon_int = e_lon_I;
    vehicle_step(cs, p, idx, l, dt, accel, steer, sin_h, cos_h, pose);
}

// ------------------------------------------------------------------------------------------------
// State.collisions() for one tile (state.py:306-310 -> state/utils.py:10-49 -> utils.py:28-62).
// Returns this lane's adjacency row (bit j = tile slot j).  lds: 9*64 doubles per wave.
// ------------------------------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ uint64_t tile_collisions(bool present, double x, double y, double s, double c,
                                                    double bw, double bl, double bcx, double bcy, int lane,
                                                    double *lds, uint64_t *mult_rows /* pre-alias row, for event multiplicity */)
{
    const int base = lane & ~(G - 1), slot = lane & (G - 1);
    // bounding circle about the box centre
    double ccx = x + (bcx * c - bcy * s), ccy = y + (bcx * s + bcy * c);
    double rad = 0.5 * __builtin_sqrt(bl * bl + bw * bw);
    uint64_t pmask = __ballot(present);
    uint64_t cand = 0;
#pragma unroll 4
    for (int j = 0; j < G; ++j) {
        double ox = shfl_d(ccx, base + j), oy = shfl_d(ccy, base + j), orad = shfl_d(rad, base + j);
        double dx = ox - ccx, dy = oy - ccy, rr = (orad + rad) * (1.0 + 1e-9) + 1e-9;
        bool hit = (dx * dx + dy * dy <= rr * rr) && ((pmask >> (base + j)) & 1) && (j != slot);
        cand |= (uint64_t)hit << j;
    }
    cand = present ? cand : 0;
    *mult_rows = 0;
    if (!__any(cand != 0)) return 0; // wave-uniform

    double A[8];
    sg_corners(x, y, s, c, bw, bl, bcx, bcy, A);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) lds[k * 64 + lane] = A[k];
    __syncthreads();
    uint64_t rows = 0, eq = 0;
    while (__any(cand != 0)) {
        if (cand) {
            int j = __builtin_ctzll(cand);
            cand &= cand - 1;
            double B[8];
            bool same = true;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                B[k] = lds[k * 64 + base + j];
                same = same && (B[k] == A[k]);
            }
            if (same) eq |= 1ull << j;                      // g == g_prime: never listed (utils.py:59)
            else if (sg_quads_intersect(A, B)) rows |= 1ull << j;
        }
    }
    *mult_rows = rows;
    if (__any(eq != 0)) { // geometry -> LAST entity owning it (state/utils.py:32-40)
        int last = 63 - __builtin_clzll(eq | (1ull << slot));
        int *li = (int *)(lds + 8 * 64);
        __syncthreads();
        li[lane] = last;
        __syncthreads();
        uint64_t nr = 0, tmp = rows;
        while (tmp) {
            int j = __builtin_ctzll(tmp);
            tmp &= tmp - 1;
            nr |= 1ull << li[base + j];
        }
        rows = nr;
    }
    return rows;
}

// ------------------------------------------------------------------------------------------------
// BatchReplayEntity.add_entities stage 1 (entity/batch.py:83-109): resample every batch-replay
// trajectory onto its scenario's union grid.  One thread per (grid row, entity slot).
// ------------------------------------------------------------------------------------------------
__global__ void build_grid_kernel(Params p, const int32_t *row_scen /*[totalN]*/, int64_t total_rows)
{
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t row = gid / p.EP;
    int e = (int)(gid - row * p.EP);
    if (row >= total_rows) return;
    int r = row_scen[row];
    size_t idx = (size_t)r * p.EP + e;
    double out[6] = {0, 0, 0, 0, 0, 0};
    if (e < p.E && p.kind[idx] == SG_KIND_REPLAY) {
        double tq = p.grid_t[row];
        const double *kn = p.knots + p.knot_off[idx] * 7;
        int n = p.knot_n[idx];
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

// ------------------------------------------------------------------------------------------------
// The rollout kernel: ScenarioGym.reset_scenario / step / rollout (scenario_gym.py:217-267) for
// 64/G scenarios per wavefront.  do_reset: State.reset first.  force: step done scenarios too
// (gym.step()); otherwise each scenario stops at is_done (gym.rollout()).
// ------------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(64, SG_WAVES_PER_SIMD) void rollout_kernel(Params p, double timestep, int n_steps, int do_reset,
                                                     int force, const double *actions /*[n][R][2]*/)
{
    __shared__ double lds[9 * 64];
    const int lane = threadIdx.x;
    const int gl = blockIdx.x * 64 + lane;
    const int r_raw = gl / G, slot = gl & (G - 1), base = lane & ~(G - 1);
    const bool in_range = r_raw < p.R;
    const int r = in_range ? r_raw : p.R - 1;
    const size_t idx = (size_t)r * p.EP + slot;
    const int kind = (in_range && slot < p.E) ? p.kind[idx] : SG_KIND_NONE;
    const int ego = p.ego[r];
    const bool is_ego = in_range && slot == ego;
    const double bw = p.bbox[0][idx], bl = p.bbox[1][idx], bcx = p.bbox[2][idx], bcy = p.bbox[3][idx];
    const double min_t = p.min_t[idx], max_t = p.max_t[idx];
    const int nk = p.knot_n[idx];
    const double *kn = p.knots + p.knot_off[idx] * 7;
    const bool is_static = nk == 1;
    const double length = p.length[r];
    const bool is_agent = kind >= SG_KIND_AGENT_REPLAY;

    Table T;
    if (kind == SG_KIND_REPLAY) {
        size_t go = (size_t)p.grid_off[r];
        T.x = p.grid_t + go; T.xs = 1;
        T.y = p.grid_y + go * 6 * p.EP + slot; T.ys = 6 * p.EP; T.cs = p.EP;
        T.n = p.grid_n[r];
    } else if (is_agent) {
        T.x = kn; T.xs = 7; T.y = kn + 1; T.ys = 7; T.cs = 1; T.n = nk;
    } else {
        T.x = nullptr; T.y = nullptr; T.n = 0; T.xs = T.ys = T.cs = 0;
    }

    double pose[6], vel[6], dist, t, prev_t;
    CtrlState cs;
    bool present;
    int done, steps;
    double m_avg, m_max, m_t, m_dist;
    uint64_t last_row, row = 0;
    int n_ev;
    double sin_h, cos_h;

    if (do_reset) {
        // ---- State.reset(t0), state.py:106-143 ----
        t = p.t0[r];
        present = false;
#pragma unroll
        for (int c = 0; c < 6; ++c) { pose[c] = 0.0; vel[c] = 0.0; }
        if (kind != SG_KIND_NONE) {
            bool inside = (t >= min_t) && (t <= max_t);
            if (is_static) { own_position_extrap(kn, nk, t, pose); present = true; }
            else if (inside) { own_position_extrap(kn, nk, t, pose); present = true; }
            else if (p.persist) { // extrapolate=(False, False): clamp
                const double *rowp = t < min_t ? kn : kn + (size_t)(nk - 1) * 7;
                for (int c = 0; c < 6; ++c) pose[c] = rowp[1 + c];
                present = true;
            }
            if (present && inside) { // Trajectory.velocity_at_t, trajectory.py:243-273
                const double eps = 1e-4;
                double a[6], b[6];
                own_position_extrap(kn, nk, t + eps / 2, a);
                own_position_extrap(kn, nk, t - eps / 2, b);
                for (int c = 0; c < 6; ++c) vel[c] = (a[c] - b[c]) / eps;
            }
        }
        prev_t = t - 0.1; // state.py:135
        dist = 0.0;
        cs.speed = present ? sg_norm2(vel[0], vel[1]) : 0.0; // controller.py:100-103
        cs.e_lon_prev = cs.e_lat_prev = cs.e_lon_int = 0.0;   // controller.py:198-203
        done = 0;
        steps = 0;
        double v0 = sg_norm3(vel[0], vel[1], vel[2]); // metrics/trajectory.py:13-17, 36-39
        m_avg = m_max = present ? v0 : __builtin_nan("");
        m_t = 0.0;
        m_dist = __builtin_nan("");
        last_row = 0;
        n_ev = 0;
    } else {
        t = p.t[r];
        prev_t = p.prev_t[r];
        present = p.present[idx] != 0;
#pragma unroll
        for (int c = 0; c < 6; ++c) { pose[c] = p.pose[c][idx]; vel[c] = p.vel[c][idx]; }
        dist = p.dist[idx];
        cs.speed = p.cs[0][idx]; cs.e_lon_prev = p.cs[1][idx];
        cs.e_lat_prev = p.cs[2][idx]; cs.e_lon_int = p.cs[3][idx];
        done = p.done[r];
        steps = p.n_steps[r];
        m_avg = p.m_avg[r]; m_max = p.m_max[r]; m_t = p.m_t[r]; m_dist = p.m_dist[r];
        last_row = p.last_row[r];
        n_ev = p.n_events[r];
    }
    sg_sincos(pose[3], sin_h, cos_h);

    Segment S;
    S.cur = seg_locate(T, t);
    seg_load(T, S);

    uint64_t mult_rows = 0;
    if (do_reset) {
        row = tile_collisions<G>(present, pose[0], pose[1], sin_h, cos_h, bw, bl, bcx, bcy, lane, lds, &mult_rows);
        if (in_range && slot < p.EP) {
#pragma unroll
            for (int c = 0; c < 6; ++c) { p.pose[c][idx] = pose[c]; p.vel[c][idx] = vel[c]; }
            p.dist[idx] = dist;
            p.coll[idx] = row;
            p.present[idx] = present;
            p.cs[0][idx] = cs.speed; p.cs[1][idx] = 0.0; p.cs[2][idx] = 0.0; p.cs[3][idx] = 0.0;
            if (p.rec_cap > 0) {
#pragma unroll
                for (int c = 0; c < 6; ++c)
                    p.rec_pose[((size_t)0 * 6 + c) * p.R * p.EP + idx] = present ? pose[c] : __builtin_nan("");
            }
            if (slot == 0) {
                p.t[r] = t; p.prev_t[r] = prev_t; p.done[r] = 0; p.n_steps[r] = 0;
                p.n_events[r] = 0; p.last_row[r] = 0;
                if (p.rec_cap > 0) { p.rec_t[r] = t; p.rec_rows[r] = 1; }
            }
            if (is_ego) { p.m_avg[r] = m_avg; p.m_max[r] = m_max; p.m_t[r] = 0.0; p.m_dist[r] = m_dist; }
        }
    }

    for (int k = 0; k < n_steps; ++k) {
        const bool run = in_range && (force || !done);
        if (!__any(run)) break;

        const double next_t = t + timestep; // scenario_gym.py:229
        const double state_dt = t - prev_t; // State.dt, state.py:198-201
        const double dt = next_t - t;       // = new State.dt after the step
        seg_advance(T, S, next_t);
        double tgt[6];
        {
            double dq = next_t - S.x_lo;
#pragma unroll
            for (int c = 0; c < 6; ++c) tgt[c] = S.sl[c] * dq + S.ylo[c];
        }

        // ---- new poses: scenario_gym.py:233-245 ----
        double np_[6];
        bool npres = false;
#pragma unroll
        for (int c = 0; c < 6; ++c) np_[c] = tgt[c];
        CtrlState ncs = cs;
        if (kind == SG_KIND_REPLAY) { // BatchReplayEntity.step, batch.py:34-53
            npres = p.persist || is_static || (next_t >= min_t && next_t <= max_t);
        } else if (is_agent) {
            if (present) {
                npres = true;
                if (kind != SG_KIND_AGENT_REPLAY) {
#pragma unroll
                    for (int c = 0; c < 6; ++c) np_[c] = pose[c];
                    if (kind == SG_KIND_AGENT_PID) {
                        pid_step(ncs, p, idx, bl, state_dt, dt, tgt[0], tgt[1], sin_h, cos_h, np_);
                    } else {
                        const double *a = actions + ((size_t)k * p.R + r) * 2;
                        double accel = actions ? a[0] : 0.0, steer = actions ? a[1] : 0.0;
                        vehicle_step(ncs, p, idx, bl, dt, accel, steer, sin_h, cos_h, np_);
                    }
                }
            } else if (min_t >= t) { // scenario_gym.py:240-244: spawn at trajectory start
                npres = true;
            }
        }

        // ---- State.update_poses / update_statistics, state.py:203-239 ----
        double prev[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) prev[c] = pose[c];
        if (npres && !present) own_position_extrap(kn, nk, t, prev); // newcomer, state.py:219-222
        double d[6], nvel[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            d[c] = np_[c] - prev[c];
            nvel[c] = d[c] / dt;
        }
        double ndist = dist + sg_norm3(d[0], d[1], d[2]);

        // commit (lanes of scenarios that are already done keep their state)
        if (run) {
            present = npres;
            cs = ncs;
            if (npres) {
#pragma unroll
                for (int c = 0; c < 6; ++c) { pose[c] = np_[c]; vel[c] = nvel[c]; }
                dist = ndist;
            }
            prev_t = t;
            t = next_t;
            ++steps;
        }
        sg_sincos(pose[3], sin_h, cos_h);

        // ---- State.collisions ----
        uint64_t nrow = tile_collisions<G>(present, pose[0], pose[1], sin_h, cos_h, bw, bl, bcx, bcy, lane, lds, &mult_rows);
        if (run) row = nrow;

        // ---- check_terminal, state.py:268-270, 397-408 ----
        int ndone = 0;
        if ((p.term_mask & SG_TERM_MAX_LENGTH) && (t + dt > length)) ndone = 1;
        uint64_t any_row = __ballot(row != 0) >> base;
        if (G < 64) any_row &= (1ull << (G & 63)) - 1;
        if ((p.term_mask & SG_TERM_COLLISION) && any_row) ndone = 1;
        uint64_t row0 = __shfl(row, base, 64);
        bool pres0 = (__ballot(present) >> base) & 1;
        if ((p.term_mask & SG_TERM_EGO_COLLISION) && pres0 && row0) ndone = 1;
        if (run) done = ndone;

        // ---- metrics, scenario_gym.py:251-252 (ego lane only) ----
        if (run && is_ego && present) {
            double speed = sg_norm3(vel[0], vel[1], vel[2]);
            double w = m_t / t; // EgoAvgSpeed._step, metrics/trajectory.py:19-24
            m_avg += (1.0 - w) * (speed - m_avg);
            m_t = t;
            m_max = __builtin_fmax(speed, m_max); // EgoMaxSpeed, :41-44
            m_dist = dist;                         // EgoDistanceTravelled, :60-62
            uint64_t fresh = row & ~last_row;      // CollisionMetric._step, metrics/collision.py:70-75
            while (fresh) {
                int j = __builtin_ctzll(fresh);
                fresh &= fresh - 1;
                int mult = 1;
                if (mult_rows != row) { // aliased geometries are listed once per owner
                    mult = 0;
                    const int *li = (const int *)(lds + 8 * 64);
                    uint64_t tmp = mult_rows;
                    while (tmp) { int q = __builtin_ctzll(tmp); tmp &= tmp - 1; mult += li[base + q] == j; }
                }
                for (int q = 0; q < mult; ++q) {
                    if (n_ev < p.ev_cap) {
                        sg_event ev;
                        ev.t = t; ev.scenario = r; ev.other = j;
                        ev.type = p.etype[(size_t)r * p.EP + j] == 0 ? -1 : 5;
                        ev.reserved = 0;
                        p.events[(size_t)r * p.ev_cap + n_ev] = ev;
                    }
                    ++n_ev;
                }
            }
            last_row = row;
        }

        // ---- step-materialised state ----
        if (run) {
#pragma unroll
            for (int c = 0; c < 6; ++c) { p.pose[c][idx] = pose[c]; p.vel[c][idx] = vel[c]; }
            p.dist[idx] = dist;
            p.coll[idx] = row;
            p.present[idx] = present;
            if (p.rec_cap > 0 && steps < p.rec_cap) {
#pragma unroll
                for (int c = 0; c < 6; ++c)
                    p.rec_pose[((size_t)steps * 6 + c) * p.R * p.EP + idx] = present ? pose[c] : __builtin_nan("");
                if (slot == 0) { p.rec_t[(size_t)steps * p.R + r] = t; p.rec_rows[r] = steps + 1; }
            }
        }
    }

    if (in_range) {
        p.cs[0][idx] = cs.speed; p.cs[1][idx] = cs.e_lon_prev;
        p.cs[2][idx] = cs.e_lat_prev; p.cs[3][idx] = cs.e_lon_int;
        if (slot == 0) { p.t[r] = t; p.prev_t[r] = prev_t; p.done[r] = done; p.n_steps[r] = steps; }
        if (is_ego) {
            p.m_avg[r] = m_avg; p.m_max[r] = m_max; p.m_t[r] = m_t; p.m_dist[r] = m_dist;
            p.last_row[r] = last_row; p.n_events[r] = n_ev;
        }
    }
}

} // namespace sg
