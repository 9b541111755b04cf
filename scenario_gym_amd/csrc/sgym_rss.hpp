// sgym_rss.hpp -- The RSSDistances callback: rss_entity, the line-test queue.
// Part of the gfx950 device code of the batched rollout engine; included by sgym_device.hpp (in order: every part builds on
// the ones before it), never on its own.
#pragma once

namespace sg {

// ------------------------------------------------------------------------------------------------
// RSSDistances.__call__ (metrics/rss/callback.py:58-128) on the current state of every scenario, + the flags RSS reads
// (metrics/rss/rss.py:70-104).  One workgroup per scenario, one thread per entity.  Ego frame: x lateral, y longitudinal;
// np.dot of 2-vectors = fma(a1, b1, a0 * b0), norm([u, v]) = sqrt(fma(v, v, u * u)) (probed); the per-entity history list
// is carried as (found, last): an "unsafe_*" entry exists / the latest "lateral" | "longitudinal" entry.  Same operation
// sequence as the oracle's sgo_rss_update.
// ------------------------------------------------------------------------------------------------
__device__ inline double rss_dot2(double a0, double a1, double b0, double b1) { return __builtin_fma(a1, b1, a0 * b0); }
#define RSS_DIV(a, b) ((a) / (b))
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
            if (!apart && sg_quads_intersect(Q, B)) {
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
            if (rss_seg_quad(Q, ax, ay, bx, by))
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

// The queued line tests of one rollout_kernel_rss launch / work item (see RssQueue): w = the queue of rollout wavefront w, whose
// lane l carries entity index w * 64 + l.  One wavefront; `ql`: its LDS.
__device__ __forceinline__ void rss_lines_block(const Params &p, RssQueueLds ql, size_t w)
{
    const int lane = threadIdx.x & 63;
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

} // namespace sg
