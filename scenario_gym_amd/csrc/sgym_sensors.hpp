// sgym_sensors.hpp -- Observation and read-out kernels: future collisions, rasters, collision classification, RSS per tick, the observe kernel, terminal flags.
// Part of the gfx950 device code of the batched rollout engine; included by sgym_device.hpp (in order: every part builds on
// the ones before it), never on its own.
#pragma once

namespace sg {

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
    __shared__ double cor[8][512]; // (one thread per entity slot of a tile: 256 threads, 512 for scenarios of more than 256)
    __shared__ double ego_pose[4]; // x, y, sin(theta), cos(theta)
    __shared__ int ego_pres;
    __shared__ int near_n;
    const int r = blockIdx.x, tid = threadIdx.x, nthr = (int)blockDim.x;
    const ScenStatic &ss = p.sstat[r];
    if (tid == 0) {
        const uint32_t idx = (uint32_t)r * p.EP + (uint32_t)ss.ego;
        const LanePtr dy(p.dyn + (size_t)(idx >> 6) * ((size_t)p.FROWS * 64), (idx & 63) * 8u);
        double s, c;
        sg_sincos(fld(dy, SG_F_POSE + 3) + 3.14159265358979311600e+00 / 2, s, c); // pose[3] + math.pi / 2
        ego_pose[0] = fld(dy, SG_F_POSE + 0); ego_pose[1] = fld(dy, SG_F_POSE + 1);
        ego_pose[2] = s; ego_pose[3] = c;
        ego_pres = fld<uint64_t>(dy, SG_F_PRESENT) != 0;
    }
    __syncthreads();
    const double ex = ego_pose[0], ey = ego_pose[1], s = ego_pose[2], c = ego_pose[3];
    const bool ego_present = ego_pres != 0;
    unsigned char *o = out + (size_t)r * stride;
    // scenarios of more entities than the workgroup has threads go tile by tile; a grid point that an earlier tile's box
    // covers stays covered (its byte is this thread's own: written and read back by the same thread)
    for (int e0 = 0; e0 < p.E || e0 == 0; e0 += nthr) {
        const int e = e0 + tid;
        if (tid == 0) near_n = 0;
        __syncthreads();
        const uint32_t idx = (uint32_t)r * p.EP + (e < p.EP ? e : 0);
        const LanePtr st(p.stat + (size_t)(idx >> 6) * (ST_COUNT * 64), (idx & 63) * 8u);
        const LanePtr dy(p.dyn + (size_t)(idx >> 6) * ((size_t)p.FROWS * 64), (idx & 63) * 8u);
        const bool present = e < p.E && fld<uint64_t>(dy, SG_F_PRESENT) != 0;
        if (present) {
            double C[8];
            const double x = fld(dy, SG_F_POSE + 0), y = fld(dy, SG_F_POSE + 1), h = fld(dy, SG_F_POSE + 3);
            double sh, ch;
            sg_sincos(h, sh, ch);
            sg_corners(x, y, sh, ch, fld(st, ST_BW), fld(st, ST_BL), fld(st, ST_BCX), fld(st, ST_BCY), C);
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
        const int nn = near_n;
        for (int q = tid; q < nw * nh; q += nthr) {
            bool hit = e0 > 0 && o[q] != 0;
            if (!hit && ego_present) {
                const int i = q / nw, j = q - i * nw;
                const double x0 = sg_linspace_at(-width / 2, width / 2, nw, j), x1 = sg_linspace_at(-height / 2, height / 2, nh, i);
                const double px = __builtin_fma(x1, -s, x0 * c) + ex, py = __builtin_fma(x1, c, x0 * s) + ey;
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
            }
            o[q] = ego_present ? (unsigned char)hit : 0; // the reference sensor needs state.poses[entity]
        }
        __syncthreads();
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
// event i of scenario r, if a table-variant launch packed it (type >= 16): the controlled ego's (and a controlled hazard's)
// pose at that step from the table rows of `tab`, the event becomes an ordinary pending one
__device__ __forceinline__ void event_take_table_pose(const Params &p, int r, int i, const double *tab, bool from_tab, int64_t ectl)
{
    sg_event &ev = p.events[(size_t)r * p.ev_cap + i];
    if (ev.type < 16) return; // not packed: recorded by another launch
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
// is scenario r's ego a lane of the controller table, and which column
__device__ __forceinline__ bool ego_table_column(const Params &p, int r, int64_t &ectl)
{
    const uint32_t eidx = (uint32_t)r * p.EP + p.sstat[r].ego;
    const LanePtr est(p.stat + (size_t)(eidx >> 6) * (ST_COUNT * 64), (eidx & 63) * 8u);
    ectl = fld<int64_t>(est, ST_CTL);
    const int ekind = (int)(fld<int64_t>(est, ST_META) & 0xff);
    return ectl >= 0 && (ekind == SG_KIND_AGENT_PID || ekind == SG_KIND_AGENT_VEHICLE);
}
#ifdef SG_UNIT_MAIN // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ __launch_bounds__(64) void event_ego_pose_kernel(Params p, TabGroups tg)
{
    const int r = blockIdx.x;
    const int n = min(p.sdyn[r].n_events, p.ev_cap);
    if (n == 0) return;
    int n_launch;
    const double *tab;
    if (!tg.pick((unsigned)(((size_t)r * p.EP) >> 6), n_launch, tab)) return; // (its group sat the launch out: nothing packed)
    int64_t ectl;
    const bool from_tab = ego_table_column(p, r, ectl);
    for (int i = threadIdx.x; i < n; i += 64) event_take_table_pose(p, r, i, tab, from_tab, ectl);
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
    // reset: 0 an update, 1 the histories of every scenario start anew, 2 those of the scenarios flagged in p.reset_mask do
    __shared__ double ego[8]; // x, y, heading, vx, vy, width, length, present
    __shared__ int s_stale;
    const int r = blockIdx.x, tid = threadIdx.x;
    const ScenStatic &ss = p.sstat[r];
    const int steps_now = p.sdyn[r].n_steps;
    const bool anew = reset == 1 || (reset == 2 && p.reset_mask[r] != 0);
    if (tid == 0) {
        const uint32_t idx = (uint32_t)r * p.EP + (uint32_t)ss.ego;
        const LanePtr st(p.stat + (size_t)(idx >> 6) * (ST_COUNT * 64), (idx & 63) * 8u);
        const LanePtr dy(p.dyn + (size_t)(idx >> 6) * ((size_t)p.FROWS * 64), (idx & 63) * 8u);
        ego[0] = fld(dy, SG_F_POSE + 0); ego[1] = fld(dy, SG_F_POSE + 1); ego[2] = fld(dy, SG_F_POSE + 3);
        ego[3] = fld(dy, SG_F_VEL + 0); ego[4] = fld(dy, SG_F_VEL + 1);
        ego[5] = fld(st, ST_BW); ego[6] = fld(st, ST_BL); ego[7] = fld<uint64_t>(dy, SG_F_PRESENT) != 0 ? 1.0 : 0.0;
        s_stale = !anew && seen[r] == steps_now;
        seen[r] = steps_now;
    }
    __syncthreads();
    if (s_stale) return;
    for (int e = tid; e < p.E; e += (int)blockDim.x) { // (scenarios of more than 512 entities: tile by tile)
        const uint32_t idx = (uint32_t)r * p.EP + (uint32_t)e;
        const LanePtr st(p.stat + (size_t)(idx >> 6) * (ST_COUNT * 64), (idx & 63) * 8u);
        const LanePtr dy(p.dyn + (size_t)(idx >> 6) * ((size_t)p.FROWS * 64), (idx & 63) * 8u);
        const bool present = fld<uint64_t>(dy, SG_F_PRESENT) != 0;
        const double hp0 = fld(dy, SG_F_POSE + 0), hp1 = fld(dy, SG_F_POSE + 1), hp3 = fld(dy, SG_F_POSE + 3);
        const double hv0 = fld(dy, SG_F_VEL + 0), hv1 = fld(dy, SG_F_VEL + 1);
        int32_t state = anew ? 0 : rss_state[idx];
        int cd = -1;
        double s_lat = __builtin_nan(""), s_long = __builtin_nan("");
        const bool skip = p.sdyn[r].t == 0.0 || ego[7] == 0.0 || e == ss.ego || !present; // callback.py:76-78
        if (!skip)
            rss_entity(ego[0], ego[1], ego[2], ego[3], ego[4], ego[5], ego[6], hp0, hp1, hp3, hv0, hv1, fld(st, ST_BW),
                       fld(st, ST_BL), fld(st, ST_BCX), fld(st, ST_BCY), state, cd, s_lat, s_long);
        rss_state[idx] = state;
        code[idx] = cd;
        safe[(size_t)idx * 2] = s_lat;
        safe[(size_t)idx * 2 + 1] = s_long;
    }
}
#endif // SG_UNIT_MAIN

// The queued line tests of one rollout_kernel_rss launch (see RssQueue): block w = the queue of rollout wavefront w, whose
// lane l carries entity index w * 64 + l.
// (tg: the blocks of that launch -- one pipeline's part of the batch, launch_rollout; else all of them)
#ifdef SG_UNIT_RSS_LINES // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ __launch_bounds__(64) void rss_lines_kernel(Params p, TabGroups tg)
{
    __shared__ RssQueue q;
    rss_lines_block(p, (RssQueueLds)&q, tg.map(blockIdx.x));
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

// TERMINAL_CONDITIONS["ego_off_road"] (state.py:401-407) for batches with pedestrian agents, whose rollout variants do not carry
// it: evaluated by a launch of its own after every step (check_terminal is the last thing a step does to `done`, so adding a
// condition afterwards is the same as having it in the list).  One thread per scenario; entities[0] is slot 0.
#ifdef SG_UNIT_MAIN
static __global__ __launch_bounds__(64) void ego_off_road_kernel(Params p)
{
    const int r = (int)blockIdx.x * 64 + (int)threadIdx.x;
    if (r >= p.R) return;
    const uint32_t idx = (uint32_t)r * p.EP;
    const LanePtr dy(p.dyn + (size_t)(idx >> 6) * ((size_t)p.FROWS * 64), (idx & 63) * 8u);
    bool off = true;
    if (fld<uint64_t>(dy, SG_F_PRESENT) != 0 && p.road) {
        const RoadIndex RI = *p.road;
        off = !(rn_layers_at(RI, RI.net_of_scen[r], SG_LAYER_DRIVEABLE, fld(dy, SG_F_POSE + 0), fld(dy, SG_F_POSE + 1)) & SG_LAYER_DRIVEABLE);
    }
    if (off) p.sdyn[r].done = 1;
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
