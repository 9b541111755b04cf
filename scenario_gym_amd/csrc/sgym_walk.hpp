// sgym_walk.hpp -- the WALKER variant of the crowd rollout (BASELINE config 5), gfx950 device code.
//
// In a crowd most pedestrians ARRIVE: PedestrianAgent._step returns speed 0, heading 0, force 0 from then on
// (pedestrian/agent.py:64-68), and after two such steps the entity's whole state -- pose (x, y, z, 0, p, r), velocity +0,
// distance, force 0, controller speed 0 -- is a fixed point of State.step (state.py:203-239): nothing about it changes any
// more except the bits other entities set in its collision row.  In the 1024 x 256 x 10,000 benchmark 68 % of all
// pedestrian-steps are such fixed points, and rollout_kernel_crowd<4> still spends a full lane on each of them.
//
// Here a scenario's workgroup holds one LANE PER ACTIVE ENTITY only (walking, or not yet at its fixed point, or not yet
// spawned, plus entity 0 and the ego): 64 or 128 lanes instead of 256.  The fixed-point entities ("statics") are rows of the
// LDS tables, written once per launch: they are neighbours and collision partners of the active lanes, never workers.
//   * social force: crowd_pair on the same LDS tables, indexed by ENTITY, candidate rows in entity bit order -- every sum
//     keeps the reference's neighbour order (social_force.py:64-84), bit for bit the arithmetic of rollout_kernel_crowd.
//   * collisions (state/utils.py:10-49): an active lane tests itself against all entities (stripe masks -> fp32 circles ->
//     fp32 SAT filter -> fp64 exact SAT, the stages of tile_collisions); the predicate is symmetric, so a hit on a static
//     is also scattered into that static's row (LDS bit per lane), and a pass over the touched statics rewrites their rows
//     in memory: base row (static-static hits, constant, saved by walk_classify_kernel) | the active hits of this step.
//   * everything a step decides is computed BEFORE anything of the step is stored.  A condition this variant does not
//     handle -- two equal geometries (utils.py:59, state/utils.py:32-40), coordinates beyond the stripe range, an operand
//     outside crowd_pair's guards -- makes the workgroup stop at the last completed step ("bail"); the host's next launch
//     (rollout_kernel_crowd with WalkSel::want = -1) finishes the chunk from the state in memory.
// The host (launch_crowd_chunks, sgym_hip.hip) cuts a rollout into chunks of steps; walk_classify_kernel sorts the
// scenarios of each chunk into classes: 0 = rollout_kernel_crowd, 1 = walk_kernel<1> (<= 64 active), 2 = walk_kernel<2>.
// Results are bit-identical to rollout_kernel_crowd<4> alone (tests: SG_CROWD_WALK=0 / 1, chunk lengths 1..256).
#pragma once
#include "sgym_device.hpp"

namespace sg {

constexpr int WALK_SLOTS = 256; // entity slots of a scenario (129..256 entities: p.WV == 4)
constexpr int WALK_NW = 4;      // 64-bit words of a row

// per-scenario scratch of the walker variant (device arrays, launch_crowd_chunks)
struct WalkArgs {
    int8_t *cls;            // [R] class of the scenario in this chunk
    int32_t *target;        // [R] steps-since-reset at which the chunk ends
    int32_t *n_active;      // [R]
    uint8_t *ent;           // [R][128] entity of active lane l (ascending), the first n_active entries
    uint64_t *smask;        // [R][4] entities at their fixed point ("statics")
    uint64_t *base;         // [R][256][4] collision row of a static restricted to statics (constant while they are static)
    int32_t *stats;         // [8] counters: scenarios per class, bails (diagnostics)
    unsigned long long *stats64; // [16] experiment builds: phase cycles
};

// Is entity e of block `blk` (64-slot state block) at the fixed point?  All from memory: the state the previous launch left.
__device__ __forceinline__ bool walk_is_static(const LanePtr &dy, const LanePtr &st, int kind, const double *routes)
{
    if (kind != SG_KIND_AGENT_PEDESTRIAN) return false;
    if (fld<uint64_t>(dy, SG_F_PRESENT) == 0) return false;
    const int64_t rt = fld<int64_t>(st, ST_ROUTE);
    const int nwp = (int)(rt >> 48);
    const int goal_idx = (int)fld(dy, SG_F_CTRL + 1);
    if (goal_idx <= nwp - 1) return false; // still walking (pedestrian/agent.py:59-62)
    uint64_t bits = 0;
#pragma unroll
    for (int c = 0; c < 6; ++c) bits |= fld<uint64_t>(dy, SG_F_VEL + c); // every velocity +0.0
    bits |= fld<uint64_t>(dy, SG_F_POSE + 3);                             // heading +0.0 (agent.py:66)
    bits |= fld<uint64_t>(dy, SG_F_FORCE + 0) | fld<uint64_t>(dy, SG_F_FORCE + 1);
    bits |= fld<uint64_t>(dy, SG_F_CTRL + 0);                             // controller speed +0.0
    // x + (+0) keeps every x except -0.0
    const uint64_t neg0 = 0x8000000000000000ull;
    const bool xz = fld<uint64_t>(dy, SG_F_POSE + 0) == neg0 || fld<uint64_t>(dy, SG_F_POSE + 1) == neg0;
    const double maxs = fld(st, ST_CTRL + SG_C_PED_MAX_SPEED);
    return bits == 0 && !xz && maxs >= 0.0; // (fmin(fmax(0, -maxs), maxs) == +0 needs maxs >= 0; NaN fails)
}

// One workgroup of 256 threads per scenario, thread = entity slot: the class of the scenario for the coming chunk, its
// active-lane list, the static mask and the statics' base rows.
#ifdef SG_UNIT_WALK
static __global__ __launch_bounds__(256) void walk_classify_kernel(Params p, WalkArgs wa, int chunk_len, int enable_mask, int walk1_max)
{
    __shared__ uint64_t s_static[4], s_active[4], s_flags;
    __shared__ float s_cx[WALK_SLOTS], s_cy[WALK_SLOTS];
    const int r = blockIdx.x, e = threadIdx.x, wave = e >> 6, lane = e & 63;
    const size_t blk = (size_t)r * 4 + wave;
    const LanePtr st(p.stat + blk * (ST_COUNT * 64), lane * 8u);
    const LanePtr dy(p.dyn + blk * ((size_t)(SG_F_COLL + 4) * 64), lane * 8u);
    const ScenStatic &ss = p.sstat[r];
    const sg_scenario_state &sd = p.sdyn[r];
    const int64_t meta = fld<int64_t>(st, ST_META);
    const int kind = e < p.E ? (int)(meta & 0xff) : SG_KIND_NONE;
    const bool stat_ = walk_is_static(dy, st, kind, p.routes);
    const bool present = kind != SG_KIND_NONE && fld<uint64_t>(dy, SG_F_PRESENT) != 0;
    // an entity that is neither static nor gone for good works: present, or still to spawn (scenario_gym.py:240-244)
    const double min_t = fld(st, ST_MIN_T);
    bool active = kind != SG_KIND_NONE && !stat_ && (present || min_t >= sd.t);
    active = active || (kind != SG_KIND_NONE && (e == 0 || e == ss.ego)); // terminal conditions look at entity 0, metrics at the ego
    const bool stat = stat_ && !active;
    // what the walker variant does not do: other kinds, a pedestrian with a head rotation or a radius outside crowd_pair's
    // guards
    bool odd = kind != SG_KIND_NONE && kind != SG_KIND_AGENT_PEDESTRIAN;
    odd = odd || (kind != SG_KIND_NONE && ((meta >> 8) & 0xff) != 1);
    if (kind == SG_KIND_AGENT_PEDESTRIAN) {
        const double rr = fld(st, ST_CTRL + SG_C_PED_RADIUS), hr = fld(st, ST_CTRL + SG_C_PED_HEAD_ROT);
        odd = odd || !(hr == 0.0 && rr > 0.0 && rr < 0x1p20);
    }
    float cx = __builtin_nanf(""), cy = cx;
    if (stat) {
        const double x = fld(dy, SG_F_POSE + 0), y = fld(dy, SG_F_POSE + 1);
        odd = odd || !(crowd_sane(x, 0x1p400) & crowd_sane(y, 0x1p400)) || !(__builtin_fabs(x) < 1.0e4 && __builtin_fabs(y) < 1.0e4);
        cx = (float)(x + fld(st, ST_BCX)); // heading 0: the box centre is the reference point + the offset
        cy = (float)(y + fld(st, ST_BCY));
    }
    s_cx[e] = cx;
    s_cy[e] = cy;
    const uint64_t bs = __ballot(stat), ba = __ballot(active), bo = __ballot(odd);
    if (e == 0) s_flags = 0;
    __syncthreads();
    if (lane == 0) {
        s_static[wave] = bs;
        s_active[wave] = ba;
        if (bo) atomicOr((unsigned long long *)&s_flags, 1ull);
    }
    __syncthreads();
    // two statics with (nearly) the same box centre could be EQUAL geometries (utils.py:59: never listed, and every third
    // entity is mapped to the last owner, state/utils.py:32-40): leave such scenarios to the full kernel
    if (stat) {
        bool twin = false;
        for (int j = 0; j < WALK_SLOTS; ++j) {
            const float dx = s_cx[j] - cx, dy_ = s_cy[j] - cy; // (non-statics: NaN, compares false)
            twin = twin || (j != e && __builtin_fabsf(dx) <= 1e-3f && __builtin_fabsf(dy_) <= 1e-3f);
        }
        if (twin) atomicOr((unsigned long long *)&s_flags, 2ull);
    }
    __syncthreads();
    int n_act = 0, before = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        n_act += __builtin_popcountll(s_active[w]);
        before += w < wave ? __builtin_popcountll(s_active[w]) : 0;
    }
    before += __builtin_popcountll(ba & ((1ull << lane) - 1));
    const bool usable = s_flags == 0 && crowd_params_ok(p.sf) && !p.ped_serial && p.rec_cap == 0;
    int cls = 0;
    if (usable && n_act <= walk1_max && (enable_mask & 1)) cls = 1; // (walk1_max <= 64: busier scenarios get two wavefronts)
    else if (usable && n_act <= 128 && (enable_mask & 2)) cls = 2;
    if (active && before < 128) wa.ent[(size_t)r * 128 + before] = (uint8_t)e;
    // base rows: the hits among statics (they stay as they are while both stay static)
    if (cls != 0) {
        uint64_t *b = wa.base + ((size_t)r * WALK_SLOTS + e) * 4;
#pragma unroll
        for (int w = 0; w < 4; ++w) b[w] = stat ? (fld<uint64_t>(dy, SG_F_COLL + w) & s_static[w]) : 0ull;
    }
    if (e == 0) {
        wa.cls[r] = (int8_t)cls;
        wa.target[r] = sd.n_steps + chunk_len;
        wa.n_active[r] = n_act;
#pragma unroll
        for (int w = 0; w < 4; ++w) wa.smask[(size_t)r * 4 + w] = s_static[w];
        if (wa.stats) atomicAdd(&wa.stats[cls], 1);
    }
}
#endif // SG_UNIT_WALK

// LDS of one walker workgroup: tables by ENTITY slot (256), queues and thresholds by LANE (64 * WVL)
template <int WVL>
struct WalkLds {
    static constexpr int NL = 64 * WVL;
    static constexpr int PAIR_CAP = 320; // pairs a wavefront can hand over to its idle lanes (4 B + 16 B each)
    float cx[WALK_SLOTS], cy[WALK_SLOTS]; // box centres (NaN: absent), SoA: the all-pairs walk reads four consecutive slots at once
    float2 sc[WALK_SLOTS];                // sin, cos of the heading
    float2 half[WALK_SLOTS];              // half length, half width
    double px[WALK_SLOTS], py[WALK_SLOTS], hd[WALK_SLOTS]; // reference point (px NaN: absent), heading
    double ox[WALK_SLOTS], oy[WALK_SLOTS], sx[WALK_SLOTS], sy[WALK_SLOTS], ss[WALK_SLOTS]; // crowd_pair's per-neighbour terms
    unsigned long long wbits[WALK_SLOTS][WVL], wprev[WALK_SLOTS][WVL]; // statics: active lanes that hit them (this / previous step)
    unsigned char lane_of[WALK_SLOTS];    // 255: the entity has no lane
    unsigned char ent_of[NL];
    uint32_t nq[8][NL];
    double r2hi[NL], r2lo[NL], rad[NL];   // radius rule of the lane's pedestrian
    int vote[4][WVL];
    int misc[8];
    double gon[128];                      // cos, sin of 2 pi i / 64 (p.gon): the 64-gon of the radius rule, read in the pair loop
    alignas(16) char pair_scratch[WVL][PAIR_CAP * 20]; // (the results of handed-over pairs are double2: 16-byte LDS accesses)
};

template <int WVL>
__device__ __forceinline__ void walk_sync()
{
    if (WVL == 1) tile_sync<1>();
    else __syncthreads();
}

// OR of two flags over the workgroup (one barrier for WVL > 1; see block_vote)
template <int WVL>
__device__ __forceinline__ int walk_vote(WalkLds<WVL> &L, int site, bool b0, bool b1 = false)
{
    const int mine = (sg_any(b0) ? 1 : 0) | (sg_any(b1) ? 2 : 0);
    if (WVL == 1) return mine;
    if ((threadIdx.x & 63) == 0) L.vote[site][threadIdx.x >> 6] = mine;
    __syncthreads();
    int r = 0;
#pragma unroll
    for (int w = 0; w < WVL; ++w) r |= L.vote[site][w];
    return r;
}

// crowd_pairs (sgym_device.hpp) for walker lanes: the neighbour sums of one wavefront, pairs balanced over its lanes, the
// candidate row (256 entity bits) walked as a queue of its non-empty 32-bit words.  Same pair arithmetic, same order of the
// sums.  `ql` = this lane's column of the per-lane tables; a pair outside crowd_pair's guards raises `bail` (the caller stops
// before the step is committed) instead of being recomputed here.
// ILP: (pedestrian, neighbour) pairs a lane evaluates per round of the loop (one LDS round trip for the operands of all of
// them).  (The pair arithmetic itself gains nothing from more chains side by side: tools/dbg/pair_bench.hip measures ~700
// cycles per pair for one chain of a wavefront that is alone on its SIMD -- 160 instructions at the fp64 issue rate -- and the
// same with 2, 4 or 8 interleaved chains, crowd_pair_n.)
template <int WVL, int ILP>
__device__ __forceinline__ void walk_pairs(const Params &p, WalkLds<WVL> &L, const CrowdConsts &C, int e, int ql,
                                           const uint64_t (&nbr)[WALK_NW], bool go, double &fx, double &fy, bool &bail, WalkTimers &wt)
{
    constexpr int CAP = WalkLds<WVL>::PAIR_CAP, ND = 2 * WALK_NW;
    const int lane = threadIdx.x & 63, wave = WVL == 1 ? 0 : (int)(threadIdx.x >> 6);
    uint32_t *list = reinterpret_cast<uint32_t *>(L.pair_scratch[wave]);
    double2 *res = reinterpret_cast<double2 *>(list + CAP);
    int n = 0, nw = 0;
    uint32_t idxs = 0;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const uint32_t d = go ? (uint32_t)(nbr[i >> 1] >> ((i & 1) * 32)) : 0u;
        if (d) {
            L.nq[nw & 7][ql] = d;
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
#ifdef SG_WALK_TIMERS
    wt.acc[1] += (unsigned long long)total; // (a count: candidate pairs of the wavefront)
#endif
    const int excess = max(n - T, 0), spare = max(T - n, 0);
    int scan = excess | (spare << 16);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int u = __shfl_up(scan, o, 64);
        if (lane >= o) scan += u;
    }
    const int listed_all = min(__shfl(scan, 63, 64) & 0xffff, CAP);
    const int e0 = (scan & 0xffff) - excess;
    const int out = min(max(CAP - e0, 0), excess);
    int h = min((scan >> 16) - spare, listed_all);
    const int h_end = min((scan >> 16), listed_all);
    const int keep = n - out;
    tile_sync<1>();
    WT(12);
    if (sg_any(out > 0)) { // hand over the LAST `out` neighbours, written in entity order
        int qe = nw - 1;
        uint32_t curh = L.nq[max(qe, 0)][ql];
        for (int q = 0; sg_any(q < out); ++q) {
            if (q < out) {
                const int bit = 31 - __builtin_clz(curh);
                const int j = (int)((idxs >> (3 * qe)) & 7u) * 32 + bit;
                list[e0 + out - 1 - q] = (uint32_t)j | ((uint32_t)lane << 8);
                curh &= ~(1u << bit);
                if (curh == 0) {
                    qe = max(qe - 1, 0);
                    curh = L.nq[qe][ql];
                }
            }
        }
    }
    tile_sync<1>();
    WT(13);
    const int wave_ql = ql - lane; // column of lane 0 of this wavefront
    int k = 0, qi = 0;
    uint32_t cur = L.nq[0][ql];
    while (sg_any((k < keep) | (h < h_end))) {
#ifdef SG_WALK_TIMERS
        wt.acc[7] += ILP; // (a count, not cycles: pair evaluations per lane)
#endif
        bool own[ILP], help[ILP], act[ILP], bad[ILP], ring[ILP];
        int jj[ILP], oq[ILP], oe[ILP], hi_[ILP];
        uint32_t ent[ILP];
        double c1x[ILP], c1y[ILP], c2x[ILP], c2y[ILP];
#pragma unroll
        for (int u = 0; u < ILP; ++u) {
            own[u] = k < keep;
            help[u] = !own[u] & (h < h_end);
            const int bit = __builtin_ctz(cur | 0x80000000u);
            const int jo = (int)((idxs >> (3 * qi)) & 7u) * 32 + bit;
            const uint32_t nxt = L.nq[min(qi + 1, 7)][ql];
            const uint32_t rest = cur & (cur - 1);
            const bool adv = own[u] & (rest == 0);
            cur = own[u] ? (adv ? nxt : rest) : cur;
            qi += adv;
            k += own[u];
            hi_[u] = min(h, CAP - 1);
            ent[u] = list[hi_[u]];
            h += help[u];
            jj[u] = own[u] ? jo : (int)(ent[u] & 0xffu);
            oq[u] = own[u] ? ql : wave_ql + (int)((ent[u] >> 8) & 63);
            oe[u] = own[u] ? e : (int)L.ent_of[oq[u]];
        }
#pragma unroll
        for (int u = 0; u < ILP; ++u) {
            const int j = jj[u], o = oe[u];
            const double rx = L.px[o] - L.px[j], ry = L.py[o] - L.py[j];
            double d2;
            crowd_pair(C, rx, ry, L.ox[j], L.oy[j], L.sx[j], L.sy[j], L.ss[j], c1x[u], c1y[u], c2x[u], c2y[u], d2, bad[u]);
            const bool valid = own[u] | help[u];
            const bool outside = d2 > L.r2hi[oq[u]], inside = d2 < L.r2lo[oq[u]];
            ring[u] = valid & !(outside | inside);
            act[u] = valid & inside;
        }
        bool any_ring = false, any_bad = false;
#pragma unroll
        for (int u = 0; u < ILP; ++u) any_ring |= ring[u];
        if (sg_any(any_ring)) { // rare: between the inscribed circle and the vertices of the 64-gon Point.buffer(r)
#pragma unroll
            for (int u = 0; u < ILP; ++u)
                if (ring[u]) act[u] = sg_in_radius(L.px[oe[u]], L.py[oe[u]], L.rad[oq[u]], L.px[jj[u]], L.py[jj[u]], L.gon);
        }
#pragma unroll
        for (int u = 0; u < ILP; ++u) any_bad |= bad[u] & act[u];
        bail = bail | any_bad; // (voted by the caller)
#pragma unroll
        for (int u = 0; u < ILP; ++u) {
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
    tile_sync<1>();
    WT(14);
    for (int q = 0; sg_any(q < out); ++q) {
        if (q < out) {
            const uint32_t en = list[e0 + q];
            const double2 c1 = res[e0 + q];
            if (!(en & (1u << 16))) {
                fx += c1.x; fy += c1.y;
                fx += (en & (1u << 17)) ? -0.0 : 0.0; fy += (en & (1u << 18)) ? -0.0 : 0.0;
            }
        }
    }
    tile_sync<1>();
    WT(15);
}

// What a lane needs to know about itself for the collision pass (constant over the launch)
struct WalkLane {
    int e;                  // entity slot, -1: idle lane
    bool is_ped_type;
    double bcx, bcy, bw, bl;
    float rad_thr, trig_eps, nbr_thr, hl, hw;
};

// State.collisions() of the active lanes against every entity + the neighbour candidates of the coming step, for the state
// (npres, x, y, h, vx, vy).  scatter: record the hits on statics in L.wbits (false for the launch's opening pass over the
// state in memory, whose rows are already there).  bail: see the file header; nothing is stored here but LDS.
template <int WVL>
__device__ __forceinline__ void walk_collisions(const Params &p, WalkLds<WVL> &L, const WalkLane &W, int r, bool npres,
                                                double x, double y, double h, double vx, double vy, double dtn,
                                                bool scatter, uint64_t (&rows)[WALK_NW],
                                                uint64_t (&nbr)[WALK_NW], bool &bail, int &n_static_hits, WalkTimers &wt)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int e = W.e;
    const bool act = e >= 0;
    const int es = act ? e : 0;
    float fs, fc;
    sg_sincos_f32(h, fs, fc);
    const float bcxf = (float)W.bcx, bcyf = (float)W.bcy;
    const float nanf_ = __builtin_nanf("");
    const bool pres = act && npres;
    const float fx = pres ? (float)x + (bcxf * fc - bcyf * fs) : nanf_;
    const float fy = pres ? (float)y + (bcxf * fs + bcyf * fc) : nanf_;
    const float mag = __builtin_fabsf(fx) + __builtin_fabsf(fy);
    const float reach = W.rad_thr + 1.9073486e-6f * mag;
    const float thr = reach * reach;
    const float nreach = W.nbr_thr + 1.9073486e-6f * mag;
    const float nthr = nreach * nreach;
    // the neighbour's terms of the repulsion, once per neighbour (tile_collisions)
    const double vmag = sg_norm2(vx, vy) + 0.0000000001;
    const double uox = vx / vmag, uoy = vy / vmag, stp = vmag * dtn;
    const double sxx = stp * uox, syy = stp * uoy;
    const bool insane = pres & !(crowd_sane(x, 0x1p400) & crowd_sane(y, 0x1p400) & crowd_sane(sxx, 0x1p20) & crowd_sane(syy, 0x1p20) &
                                 (stp < 0x1p20));
    walk_sync<WVL>(); // the readers of the previous pass are through
    if (act) {
        L.cx[e] = fx;
        L.cy[e] = fy;
        L.sc[e] = make_float2(fs, fc);
        L.px[e] = (pres && W.is_ped_type) ? x : __builtin_nan("");
        L.py[e] = y;
        L.hd[e] = h;
        L.ox[e] = uox; L.oy[e] = uoy;
        L.sx[e] = sxx; L.sy[e] = syy; L.ss[e] = stp * stp;
    }
    // the statics' step * step: |v| + 1e-10 with v = 0, times THIS step's dt (social_force.py:148-155)
    {
        const double st0 = 0.0000000001 * dtn, ss0 = st0 * st0;
#pragma unroll
        for (int k = 0; k < WALK_SLOTS / WalkLds<WVL>::NL; ++k) {
            const int s = tid + k * WalkLds<WVL>::NL;
            if (L.lane_of[s] == 255) L.ss[s] = ss0;
        }
    }
#pragma unroll
    for (int w = 0; w < WALK_NW; ++w) { rows[w] = 0; nbr[w] = 0; }
    WT(8);
    if (walk_vote<WVL>(L, 0, insane)) { bail = true; return; } // (uniform; the vote also publishes the LDS rows above)
    if (WVL == 1) walk_sync<WVL>();
    // ---- broad phase: every lane against all 256 slots, wave-uniform LDS broadcast reads of four slots at a time, packed
    // fp32 (tile_collisions, all_pairs): a fixed ~1.4 k instructions without a single dependent LDS round trip -- the walker
    // wavefront is alone on its SIMD, the per-candidate loops of the stripe-mask variant ran at the LDS latency.  No
    // coordinate range to respect either.  Absent slots hold NaN centres: whatever their sign bit says, the filter and the
    // pair loop drop them (NaN-safe compares), as in the wide tiles of tile_collisions. ----
    uint64_t close[WALK_NW];
    bool any_cand = false;
    {
        const v2f fx2 = {fx, fx}, fy2 = {fy, fy}, thr2 = {thr, thr}, nthr2 = {nthr, nthr};
        constexpr int TS = WALK_SLOTS, NW32 = TS / 32, PER = 8;
        uint32_t out_w[NW32], nout_w[NW32]; // bit j = 1: slot j is OUTSIDE this lane's reach
        v4f xs = *reinterpret_cast<const v4f *>(&L.cx[TS - 4]);
        v4f ys = *reinterpret_cast<const v4f *>(&L.cy[TS - 4]);
        v4f xs1 = *reinterpret_cast<const v4f *>(&L.cx[TS - 8]);
        v4f ys1 = *reinterpret_cast<const v4f *>(&L.cy[TS - 8]);
#pragma unroll
        for (int w2 = NW32 - 1; w2 >= 0; --w2) {
            uint32_t w = 0u, v = 0u;
#pragma unroll 1 // (a real loop: unrolled, the 128 LDS reads of the walk are hoisted to its top and take hundreds of registers)
            for (int q = PER - 1; q >= 0; --q) {
                const int jb = w2 * 32 + q * 4;
                v4f xs2 = xs1, ys2 = ys1; // two groups of four slots stay in flight
                if (jb >= 8) {
                    xs2 = *reinterpret_cast<const v4f *>(&L.cx[jb - 8]);
                    ys2 = *reinterpret_cast<const v4f *>(&L.cy[jb - 8]);
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
                v2f na = nthr2 - d2a, nb = nthr2 - d2b;
                v = __builtin_amdgcn_alignbit(v, __float_as_uint(nb.y), 31);
                v = __builtin_amdgcn_alignbit(v, __float_as_uint(nb.x), 31);
                v = __builtin_amdgcn_alignbit(v, __float_as_uint(na.y), 31);
                v = __builtin_amdgcn_alignbit(v, __float_as_uint(na.x), 31);
                xs = xs1; ys = ys1;
                xs1 = xs2; ys1 = ys2;
            }
            out_w[w2] = w;
            nout_w[w2] = v;
            __builtin_amdgcn_sched_barrier(0); // (else every LDS read of the walk is hoisted to its top: hundreds of registers)
        }
#pragma unroll
        for (int w = 0; w < WALK_NW; ++w) {
            uint64_t inside = ~(((uint64_t)out_w[2 * w + 1] << 32) | out_w[2 * w]);
            uint64_t nin = ~(((uint64_t)nout_w[2 * w + 1] << 32) | nout_w[2 * w]);
            if ((es >> 6) == w) { inside &= ~(1ull << (es & 63)); nin &= ~(1ull << (es & 63)); } // not with itself
            close[w] = pres ? inside : 0;
            nbr[w] = (pres && W.nbr_thr > 0.0f) ? nin : 0;
            any_cand = any_cand || close[w] != 0;
        }
    }
    WT(9);
    // ---- fp32 SAT filter (tile_collisions) ----
    uint64_t fuzzy[WALK_NW];
#pragma unroll
    for (int w = 0; w < WALK_NW; ++w) fuzzy[w] = 0;
    bool any_fuzzy = false;
    if (sg_any(any_cand)) {
#pragma unroll
        for (int w = 0; w < WALK_NW; ++w) {
            uint64_t cand = close[w];
            while (sg_any(cand != 0)) {
                if (cand) {
                    const int jl = __builtin_ctzll(cand);
                    cand &= cand - 1;
                    const int j = w * 64 + jl;
                    const float2 oc = make_float2(L.cx[j], L.cy[j]), os = L.sc[j], oh = L.half[j];
                    const float dx = oc.x - fx, dy = oc.y - fy;
                    const float cd = __builtin_fabsf(fc * os.y + fs * os.x);
                    const float sd = __builtin_fabsf(fs * os.y - fc * os.x);
                    const float g0 = __builtin_fabsf(dx * fc + dy * fs) - (W.hl + oh.x * cd + oh.y * sd);
                    const float g1 = __builtin_fabsf(dy * fc - dx * fs) - (W.hw + oh.x * sd + oh.y * cd);
                    const float g2 = __builtin_fabsf(dx * os.y + dy * os.x) - (oh.x + W.hl * cd + W.hw * sd);
                    const float g3 = __builtin_fabsf(dy * os.y - dx * os.x) - (oh.y + W.hl * sd + W.hw * cd);
                    const float gap = __builtin_fmaxf(__builtin_fmaxf(g0, g1), __builtin_fmaxf(g2, g3));
                    // the margin of the pair: the OTHER side of it would use its own trig_eps -- take a bound on both (every
                    // pedestrian of a crowd launch has the same box class; the larger of the two margins is conservative)
                    const float eps = 1e-3f + 1.9073486e-6f * (mag + __builtin_fabsf(oc.x) + __builtin_fabsf(oc.y)) + W.trig_eps;
                    bool unsure = (gap <= eps) && (gap >= -eps);
                    unsure = unsure || (dx == 0.0f && dy == 0.0f);
                    if (unsure) fuzzy[w] |= 1ull << jl;
                    else if (gap < -eps) rows[w] |= 1ull << jl;
                }
            }
            any_fuzzy = any_fuzzy || fuzzy[w] != 0;
        }
    }
    WT(10);
    // ---- exact fp64 SAT on the pairs inside the margin; the partner's corners from its LDS pose + its static rows ----
    bool eq = false;
    if (sg_any(any_fuzzy)) {
        double A[8];
        {
            double s, c;
            sg_sincos(h, s, c);
            sg_corners(x, y, s, c, W.bw, W.bl, W.bcx, W.bcy, A);
        }
#pragma unroll
        for (int w = 0; w < WALK_NW; ++w) {
            while (sg_any(fuzzy[w] != 0)) {
                if (fuzzy[w]) {
                    const int jl = __builtin_ctzll(fuzzy[w]);
                    fuzzy[w] &= fuzzy[w] - 1;
                    const int j = w * 64 + jl;
                    const double *sb = p.stat + ((size_t)r * 4 + (j >> 6)) * (ST_COUNT * 64) + (j & 63);
                    double B[8], s, c;
                    sg_sincos(L.hd[j], s, c);
                    sg_corners(L.px[j], L.py[j], s, c, sb[ST_BW * 64], sb[ST_BL * 64], sb[ST_BCX * 64], sb[ST_BCY * 64], B);
                    bool same = true;
#pragma unroll
                    for (int k = 0; k < 8; ++k) same = same && (B[k] == A[k]);
                    if (same) eq = true; // g == g_prime (utils.py:59) and the owner mapping: the full kernel's business
                    else if (sg_quads_intersect(A, B)) rows[w] |= 1ull << jl;
                }
            }
        }
    }
    WT(11);
    if (sg_any(eq)) bail = true; // (made uniform by the caller's vote)
    // ---- the same hits from the statics' side ----
    n_static_hits = 0;
    if (scatter) {
#pragma unroll
        for (int w = 0; w < WALK_NW; ++w) {
            uint64_t m = rows[w];
            while (m) {
                const int j = w * 64 + __builtin_ctzll(m);
                m &= m - 1;
                if (L.lane_of[j] == 255) {
                    atomicOr(&L.wbits[j][WVL == 1 ? 0 : (tid >> 6)], 1ull << lane);
                    ++n_static_hits;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The walker kernel.  One workgroup of 64 * WVL lanes per scenario of class WVL.
// ------------------------------------------------------------------------------------------------
template <int WVL>
__device__ __forceinline__ void walk_body(const Params &p, double timestep, int n_steps, int force, const WalkArgs &wa)
{
    using LDS = WalkLds<WVL>;
    constexpr int NL = LDS::NL;
    __shared__ LDS L;
    const int r = blockIdx.x;
    if (wa.cls[r] != WVL) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int step_target = wa.target[r];
    const int n_act = wa.n_active[r];
    const int e_raw = tid < 128 ? (int)wa.ent[(size_t)r * 128 + tid] : 0;
    const bool act = tid < n_act; // (every value 0..255 of the list is an entity: the count says where it ends)
    const int e = act ? e_raw : -1, es = act ? e_raw : 0;
    const ScenStatic &ss = p.sstat[r];
    sg_scenario_state &sd = p.sdyn[r];
    const size_t blk = (size_t)r * 4 + (es >> 6);
    const LanePtr st(p.stat + blk * (ST_COUNT * 64), (es & 63) * 8u);
    const LanePtr dy(p.dyn + blk * ((size_t)(SG_F_COLL + 4) * 64), (es & 63) * 8u);
    const int64_t meta = fld<int64_t>(st, ST_META);
    const int kind = act ? (int)(meta & 0xff) : SG_KIND_NONE;
    const bool is_ped = kind == SG_KIND_AGENT_PEDESTRIAN;
    const bool is_ego = act && e == ss.ego;
    const double min_t = fld(st, ST_MIN_T);
    const double length = ss.length;
    WalkLane W;
    W.e = e;
    W.is_ped_type = ((meta >> 8) & 0xff) == 1;
    W.bcx = fld(st, ST_BCX); W.bcy = fld(st, ST_BCY); W.bw = fld(st, ST_BW); W.bl = fld(st, ST_BL);
    const double vdes_c = fld(st, ST_CTRL + SG_C_PED_SPEED_DESIRED), maxs_c = fld(st, ST_CTRL + SG_C_PED_MAX_SPEED);
    const double rad_c = fld(st, ST_CTRL + SG_C_PED_RADIUS);
    const double *wp = nullptr;
    int nwp = 0;
    if (is_ped) {
        const int64_t rt = fld<int64_t>(st, ST_ROUTE);
        wp = p.routes + (rt & 0xffffffffffffll) * 2;
        nwp = (int)(rt >> 48);
    }
    // ---- LDS tables of all 256 entities from the state in memory ----
    const uint64_t *smask = wa.smask + (size_t)r * 4;
    float rmax = 0.0f, omax = 0.0f, omax_ped = 0.0f; // largest bounding-circle radius / centre offset of the scenario (rollout_body)
    for (int k = 0; k < WALK_SLOTS / NL; ++k) {
        const int s = tid + k * NL;
        const size_t b2 = (size_t)r * 4 + (s >> 6);
        const LanePtr st2(p.stat + b2 * (ST_COUNT * 64), (s & 63) * 8u);
        const LanePtr dy2(p.dyn + b2 * ((size_t)(SG_F_COLL + 4) * 64), (s & 63) * 8u);
        const int64_t m2 = fld<int64_t>(st2, ST_META);
        const int k2 = s < p.E ? (int)(m2 & 0xff) : SG_KIND_NONE;
        const bool stat = (smask[s >> 6] >> (s & 63)) & 1;
        const double bw = fld(st2, ST_BW), bl = fld(st2, ST_BL), bcx = fld(st2, ST_BCX), bcy = fld(st2, ST_BCY);
        if (k2 != SG_KIND_NONE || true) { // (padding slots: zero boxes, as in rollout_body's reductions)
            const float rad = (float)(0.5 * __builtin_sqrt(bl * bl + bw * bw)) * 1.000001f;
            const float off = (float)__builtin_sqrt(bcx * bcx + bcy * bcy) * 1.000001f;
            rmax = __builtin_fmaxf(rmax, rad);
            omax = __builtin_fmaxf(omax, off);
            if (((m2 >> 8) & 0xff) == 1) omax_ped = __builtin_fmaxf(omax_ped, off);
        }
        L.half[s] = make_float2((float)(0.5 * bl), (float)(0.5 * bw));
        L.lane_of[s] = 255;
        // a static: heading 0, velocity 0; everything else (absent, padding, active): filled / overwritten by its lane below
        float cx = __builtin_nanf(""), cy = cx;
        double px = __builtin_nan(""), py = 0.0;
        if (stat) {
            px = fld(dy2, SG_F_POSE + 0);
            py = fld(dy2, SG_F_POSE + 1);
            float fs, fc;
            sg_sincos_f32(0.0, fs, fc);
            cx = (float)px + ((float)bcx * fc - (float)bcy * fs);
            cy = (float)py + ((float)bcx * fs + (float)bcy * fc);
            L.sc[s] = make_float2(fs, fc);
        } else {
            L.sc[s] = make_float2(0.0f, 1.0f);
        }
        L.cx[s] = cx; L.cy[s] = cy;
        L.px[s] = px; L.py[s] = py; L.hd[s] = 0.0;
        // unit velocity 0 / (0 + 1e-10) = +0, step * 0 = +0 (tile_collisions with v = 0); ss is refreshed every step
        L.ox[s] = 0.0; L.oy[s] = 0.0; L.sx[s] = 0.0; L.sy[s] = 0.0; L.ss[s] = 0.0;
#pragma unroll
        for (int w = 0; w < WVL; ++w) {
            L.wbits[s][w] = 0;
            // hits of non-statics that the row in memory still holds: rewritten by the first step (whatever it finds)
            uint64_t other = 0;
            if (stat) {
#pragma unroll
                for (int q = 0; q < 4; ++q) other |= fld<uint64_t>(dy2, SG_F_COLL + q) & ~smask[q];
            }
            L.wprev[s][w] = other ? ~0ull : 0ull;
        }
    }
    // workgroup maxima (every lane needs them for its reach)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        rmax = __builtin_fmaxf(rmax, __shfl_xor(rmax, o, 64));
        omax = __builtin_fmaxf(omax, __shfl_xor(omax, o, 64));
        omax_ped = __builtin_fmaxf(omax_ped, __shfl_xor(omax_ped, o, 64));
    }
    if (WVL > 1) {
        float *red = reinterpret_cast<float *>(L.pair_scratch[0]);
        __syncthreads();
        if (lane == 0) { red[wave] = rmax; red[8 + wave] = omax; red[16 + wave] = omax_ped; }
        __syncthreads();
        for (int w = 0; w < WVL; ++w) {
            rmax = __builtin_fmaxf(rmax, red[w]); omax = __builtin_fmaxf(omax, red[8 + w]);
            omax_ped = __builtin_fmaxf(omax_ped, red[16 + w]);
        }
        __syncthreads();
    }
    {
        const float rad = (float)(0.5 * __builtin_sqrt(W.bl * W.bl + W.bw * W.bw)) * 1.000001f;
        const float off = (float)__builtin_sqrt(W.bcx * W.bcx + W.bcy * W.bcy) * 1.000001f;
        W.rad_thr = rad + rmax + 2e-3f + 2.0f * SG_TRIG32_ERR * (off + omax);
        W.trig_eps = SG_TRIG32_ERR * (12.0f * W.rad_thr + 4.0f * (off + omax));
        W.nbr_thr = is_ped ? (float)rad_c * 1.000001f + off + omax_ped + 2e-3f + 2.0f * SG_TRIG32_ERR * (off + omax_ped) : 0.0f;
        W.hl = (float)(0.5 * W.bl);
        W.hw = (float)(0.5 * W.bw);
    }
    for (int k = tid; k < 128; k += NL) L.gon[k] = p.gon[k];
    walk_sync<WVL>();
    if (act) {
        L.lane_of[e] = (unsigned char)tid;
        L.ent_of[tid] = (unsigned char)e;
        const double r2 = rad_c * rad_c;
        L.r2hi[tid] = r2 * (1.0 + 1e-9);
        L.r2lo[tid] = r2 * 0.9975;
        L.rad[tid] = rad_c;
    } else {
        L.ent_of[tid] = 0;
        L.r2hi[tid] = 0.0; L.r2lo[tid] = 0.0; L.rad[tid] = 0.0;
    }
    walk_sync<WVL>();
    bool bail = false;
    // ---- the lane's state from memory (rollout_body, continuing launch) ----
    CrowdConsts CC{};
    {
        const RecipDiv rs(p.sf.ped_repulse_sigma);
        CC.k2_scale = p.sf.ped_repulse_V / p.sf.ped_repulse_sigma;
        CC.sig_b = rs.b;
        CC.sig_r = rs.r;
        CC.cos_sight = p.sf.cos_sight;
        CC.sight_weight = p.sf.sight_weight;
        CC.k3 = 2 * p.sf.ped_attract_C;
    }
    double pose[6], dist, t = sd.t, prev_t = sd.prev_t;
    bool present = act && fld<uint64_t>(dy, SG_F_PRESENT) != 0;
#pragma unroll
    for (int c = 0; c < 6; ++c) pose[c] = fld(dy, SG_F_POSE + c);
    double velx = fld(dy, SG_F_VEL + 0), vely = fld(dy, SG_F_VEL + 1);
    dist = fld(dy, SG_F_DIST);
    double cspeed = fld(dy, SG_F_CTRL + 0);
    int goal_idx = (int)fld(dy, SG_F_CTRL + 1);
    const double ctrl2 = fld(dy, SG_F_CTRL + 2), ctrl3 = fld(dy, SG_F_CTRL + 3);
    double m_avg = sd.ego_avg_speed, m_max = sd.ego_max_speed, m_t = sd.avg_t;
    uint64_t last_row[WALK_NW], row[WALK_NW], nbr[WALK_NW];
#pragma unroll
    for (int w = 0; w < WALK_NW; ++w) { last_row[w] = sd.last_row[w]; row[w] = fld<uint64_t>(dy, SG_F_COLL + w); }
    int n_ev = sd.n_events, done = sd.done, steps = sd.n_steps;
    long long noise_pos = sd.noise_pos;
    const bool base_any = [&] { // some static-static hit exists (terminal condition "collision")
        bool any = false;
        for (int k = 0; k < WALK_SLOTS / NL; ++k) {
            const uint64_t *b = wa.base + ((size_t)r * WALK_SLOTS + tid + k * NL) * 4;
            any = any || (b[0] | b[1] | b[2] | b[3]) != 0;
        }
        return walk_vote<WVL>(L, 3, any) != 0;
    }();
    sg_loads_done();
    // ---- opening pass: LDS entries, stripe membership and neighbour candidates of the state in memory ----
    WalkTimers wt;
#ifdef SG_WALK_TIMERS
    for (int i = 0; i < 16; ++i) wt.acc[i] = 0;
    wt.last = __builtin_amdgcn_s_memtime();
#endif
    int n_hits = 0;
    {
        uint64_t tmp[WALK_NW];
        walk_collisions<WVL>(p, L, W, r, present, pose[0], pose[1], pose[3], velx, vely, (t + timestep) - t, false,
                             tmp, nbr, bail, n_hits, wt);
        bail = walk_vote<WVL>(L, 1, bail) != 0;
    }
    const double *Kp = SG_TRIG;
#ifdef SG_WALK_TIMERS
    wt.last = __builtin_amdgcn_s_memtime();
#endif
    bool first_store = true; // this launch has not stored its velocity z / pitch / roll rows yet
    for (int k = 0; k < n_steps && !bail; ++k) {
        if (!((force || !done) && steps < step_target)) break; // (uniform: one scenario per workgroup)
        asm volatile("" : "+s"(Kp));
        ConstTbl K = (ConstTbl)Kp;
        const double next_t = t + timestep; // scenario_gym.py:229
        const double state_dt = t - prev_t;
        const double dt = next_t - t;
        // ---- PedestrianAgent.step, part 1: the social force (ped_force, CROWD) ----
        bool go = false;
        double fx = 0.0, fy = 0.0, vdes = 0.0;
        if (is_ped && present) {
            if (goal_idx <= nwp - 1) goal_idx = ped_goal_update(wp, nwp, pose[0], pose[1]);
            if (goal_idx <= nwp - 1) {
                go = true;
                double gx = wp[2 * goal_idx] - pose[0], gy = wp[2 * goal_idx + 1] - pose[1]; // _force_to_goal, :119-138
                double gn = sg_norm2(gx, gy);
                if (gn == 0) gn += 0.000000001;
                vdes = vdes_c;
                const double inv_tau = 1 / p.sf.relaxation_time;
                fx = inv_tau * (vdes * (gx / gn) - velx);
                fy = inv_tau * (vdes * (gy / gn) - vely);
            }
        }
        bool pbail = false;
        WT(0);
        walk_pairs<WVL, 2>(p, L, CC, es, tid, nbr, go, fx, fy, pbail, wt);
        WT(1);
        // ---- random fluctuations (rollout_body) ----
        PedNoise nz{0.0, 0.0, false};
        if (p.noise_mode == 1) {
            const uint64_t walk = __ballot(go);
            int before = __builtin_popcountll(walk & ((1ull << lane) - 1)), count = __builtin_popcountll(walk);
            if (WVL > 1) { // walkers in the wavefronts before this one
                if (lane == 0) L.misc[wave] = count;
                __syncthreads();
                count = 0;
                for (int w = 0; w < WVL; ++w) {
                    const int c = L.misc[w];
                    before += w < wave ? c : 0;
                    count += c;
                }
                __syncthreads();
            }
            const long long at = noise_pos + 2 * before;
            if (go) {
                const bool inside = at + 1 < p.noise_len;
                const double *z = p.noise_normals + (size_t)r * (size_t)p.noise_len + (inside ? at : 0);
                nz = PedNoise{p.noise_std_lon * (inside ? z[0] : 0.0), p.noise_std_lat * (inside ? z[1] : 0.0), true};
            }
            noise_pos += 2 * count;
        } else if (p.noise_mode == 2) {
            double z0, z1;
            sg_noise_pair(p.noise_seed, (uint32_t)r, (uint32_t)es, (uint32_t)steps, z0, z1, K);
            nz = PedNoise{p.noise_std_lon * z0, p.noise_std_lat * z1, true};
        }
        // ---- new poses: scenario_gym.py:233-245 ----
        bool npres = false;
        double np_[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, fpx = 0.0, fpy = 0.0, ncspeed = cspeed;
        if (is_ped) {
            if (present) {
                npres = true;
                ped_move<false>(p, go, fx, fy, vdes, maxs_c, pose, state_dt, ncspeed, fpx, fpy, np_, K, nz);
            } else if (min_t >= t) { // spawn at the trajectory position of next_t
                npres = true;
                Table T = lane_table(p, SG_KIND_AGENT_PEDESTRIAN, ss, es, st);
                Segment S2;
                S2.cur = seg_locate(T, next_t);
                seg_load(T, S2);
                const double dq = next_t - S2.x_lo;
#pragma unroll
                for (int c = 0; c < 6; ++c) np_[c] = S2.sl[c] * dq + S2.ylo[c];
            }
        }
        // ---- State.update_poses / update_statistics, state.py:203-239 ----
        double d[6], vel[6];
        if (npres && !present) {
            double prev[6];
            own_position_extrap(p.knots + fld<int64_t>(st, ST_KNOT_OFF) * 7, (int)(fld<int64_t>(st, ST_META) >> 32), t, prev);
#pragma unroll
            for (int c = 0; c < 6; ++c) d[c] = np_[c] - prev[c];
        } else {
#pragma unroll
            for (int c = 0; c < 6; ++c) d[c] = np_[c] - pose[c];
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) vel[c] = d[c] / dt;
        WT(2);
        // ---- State.collisions of the new state; nothing of this step has been stored yet ----
        uint64_t nrow[WALK_NW], nnbr[WALK_NW];
        bool cbail = pbail;
        walk_collisions<WVL>(p, L, W, r, npres, np_[0], np_[1], np_[3], vel[0], vel[1], (next_t + timestep) - next_t,
                             true, nrow, nnbr, cbail, n_hits, wt);
        if (walk_vote<WVL>(L, 1, cbail)) { bail = true; break; }
        WT(3);
        // ---- commit ----
        const bool was_present = present;
        (void)was_present;
        present = npres;
        cspeed = ncspeed;
        if (npres) {
#pragma unroll
            for (int c = 0; c < 6; ++c) pose[c] = np_[c];
            dist += sg_norm3(d[0], d[1], d[2]);
            velx = vel[0];
            vely = vel[1];
        }
        prev_t = t;
        t = next_t;
        ++steps;
#pragma unroll
        for (int w = 0; w < WALK_NW; ++w) { row[w] = nrow[w]; nbr[w] = nnbr[w]; }
        if (act) {
            stf(dy, SG_F_POSE + 0, pose[0]); stf(dy, SG_F_POSE + 1, pose[1]); stf(dy, SG_F_POSE + 3, pose[3]);
            stf(dy, SG_F_POSE + 2, pose[2]); stf(dy, SG_F_POSE + 4, pose[4]); stf(dy, SG_F_POSE + 5, pose[5]);
            if (present) {
                stf(dy, SG_F_VEL + 0, vel[0]); stf(dy, SG_F_VEL + 1, vel[1]); stf(dy, SG_F_VEL + 3, vel[3]);
                stf(dy, SG_F_VEL + 2, vel[2]); stf(dy, SG_F_VEL + 4, vel[4]); stf(dy, SG_F_VEL + 5, vel[5]);
            }
            stf(dy, SG_F_DIST, dist);
            stf(dy, SG_F_PRESENT, (uint64_t)present);
            if (is_ped) { stf(dy, SG_F_FORCE + 0, fpx); stf(dy, SG_F_FORCE + 1, fpy); }
#pragma unroll
            for (int w = 0; w < WALK_NW; ++w) stf(dy, SG_F_COLL + w, row[w]);
        }
        first_store = false;
        WT(4);
        // ---- the rows of the statics the active lanes touch (or touched in the previous step) ----
        walk_sync<WVL>();
        for (int q = 0; q < WALK_SLOTS / NL; ++q) {
            const int s = tid + q * NL;
            uint64_t wnow[WVL], any = 0;
#pragma unroll
            for (int w = 0; w < WVL; ++w) { wnow[w] = L.wbits[s][w]; any |= wnow[w] | L.wprev[s][w]; }
            if (any && L.lane_of[s] == 255) {
                const uint64_t *b = wa.base + ((size_t)r * WALK_SLOTS + s) * 4;
                uint64_t rw[4] = {b[0], b[1], b[2], b[3]};
#pragma unroll
                for (int w = 0; w < WVL; ++w) {
                    uint64_t m = wnow[w];
                    while (m) {
                        const int l = __builtin_ctzll(m);
                        m &= m - 1;
                        const int o = L.ent_of[w * 64 + l];
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                            if ((o >> 6) == v) rw[v] |= 1ull << (o & 63);
                    }
                    L.wprev[s][w] = wnow[w];
                    L.wbits[s][w] = 0;
                }
                double *drow = p.dyn + ((size_t)r * 4 + (s >> 6)) * ((size_t)(SG_F_COLL + 4) * 64) + (s & 63);
#pragma unroll
                for (int v = 0; v < 4; ++v) reinterpret_cast<uint64_t *>(drow)[(SG_F_COLL + v) * 64] = rw[v];
            }
        }
        WT(5);
        // ---- ego metrics, scenario_gym.py:251-252 ----
        if (is_ego && present) {
            const double speed = sg_norm3(vel[0], vel[1], vel[2]);
            const double w = m_t / t; // EgoAvgSpeed._step, metrics/trajectory.py:19-24
            m_avg += (1.0 - w) * (speed - m_avg);
            m_t = t;
            m_max = __builtin_fmax(speed, m_max);
        }
        // ---- check_terminal, state.py:268-270, 397-408 ----
        int ndone = 0;
        if ((p.term_mask & SG_TERM_MAX_LENGTH) && (t + dt > length)) ndone = 1;
        if (p.term_mask & (SG_TERM_COLLISION | SG_TERM_EGO_COLLISION)) {
            const bool any_mine = act && (row[0] | row[1] | row[2] | row[3]) != 0;
            const int v = walk_vote<WVL>(L, 2, any_mine, act && e == 0 && present && any_mine);
            if ((p.term_mask & SG_TERM_COLLISION) && ((v & 1) || base_any)) ndone = 1;
            if ((p.term_mask & SG_TERM_EGO_COLLISION) && (v & 2)) ndone = 1;
        }
        done = ndone;
        // ---- CollisionMetric._step, metrics/collision.py:70-75 (ego lane) ----
        if (is_ego && present) {
#pragma unroll
            for (int w = 0; w < WALK_NW; ++w) {
                uint64_t fresh = row[w] & ~last_row[w];
                while (fresh) {
                    const int j = w * 64 + __builtin_ctzll(fresh);
                    fresh &= fresh - 1;
                    const double *oblk = p.stat + ((size_t)r * 4 + (j >> 6)) * (ST_COUNT * 64);
                    const int64_t ometa = reinterpret_cast<const int64_t *>(oblk)[ST_META * 64 + (j & 63)];
                    if (n_ev < p.ev_cap) {
                        sg_event *dst = &p.events[(size_t)r * p.ev_cap + n_ev];
                        struct { double t; int32_t scenario, other, type, reserved; } head;
                        head.t = t; head.scenario = (int32_t)r; head.other = j;
                        head.type = ((ometa >> 8) & 0xff) == 0 ? -1 : 5;
                        head.reserved = 0;
                        *reinterpret_cast<decltype(head) *>(dst) = head;
                        double *hp = p.ev_hpose + ((size_t)r * p.ev_cap + n_ev) * 3;
                        hp[0] = hp[1] = hp[2] = __builtin_nan("");
                        double *ep = p.ev_pose + ((size_t)r * p.ev_cap + n_ev) * 3;
                        ep[0] = pose[0]; ep[1] = pose[1]; ep[2] = pose[3];
                    }
                    ++n_ev;
                }
                last_row[w] = row[w];
            }
        }
        WT(6);
    }
#ifdef SG_WALK_TIMERS
    if (lane == 0 && wa.stats64) for (int i = 0; i < 16; ++i) if (wt.acc[i]) atomicAdd(wa.stats64 + i, wt.acc[i]);
#endif
    (void)first_store;
    // ---- write back what lives in registers ----
    if (act) {
        stf(dy, SG_F_CTRL + 0, cspeed);
        stf(dy, SG_F_CTRL + 1, is_ped ? (double)goal_idx : fld(dy, SG_F_CTRL + 1));
        stf(dy, SG_F_CTRL + 2, ctrl2);
        stf(dy, SG_F_CTRL + 3, ctrl3);
        if (e == 0) { sd.t = t; sd.prev_t = prev_t; sd.done = done; sd.n_steps = steps; sd.noise_pos = noise_pos; }
        if (is_ego) {
            sd.ego_avg_speed = m_avg; sd.ego_max_speed = m_max; sd.avg_t = m_t;
            if (steps > 0 && present) sd.ego_distance_travelled = dist;
#pragma unroll
            for (int w = 0; w < WALK_NW; ++w) sd.last_row[w] = last_row[w];
            sd.n_events = n_ev;
        }
    }
    if (tid == 0 && wa.stats && bail) atomicAdd(&wa.stats[4], 1);
}

#ifdef SG_UNIT_WALK
template <int WVL>
__global__ __launch_bounds__(64 * WVL, 1) void walk_kernel(Params p, double timestep, int n_steps, int force, WalkArgs wa)
{
    walk_body<WVL>(p, timestep, n_steps, force, wa);
}
#endif

} // namespace sg
