// sgym_walk4.hpp -- the walker step of a crowd scenario spread over the four SIMDs of a compute unit (BASELINE config 5).
//
// sgym_walk.hpp keeps one lane per ACTIVE entity and the arrived pedestrians ("statics": fixed points of State.step,
// pedestrian/agent.py:64-68) as LDS rows -- and runs that on ONE wavefront per scenario: ~15 k instructions per scenario-step,
// one wavefront per SIMD, latency-bound at ~50 us per step however few pedestrians still walk (HISTORY.md, round 4).  Here the
// same step runs on a workgroup of FOUR wavefronts that all carry the same <= 64 walker lanes and divide the 256 entity SLOTS
// between them: wavefront v searches, filters and evaluates pairs against slots 64 v .. 64 v + 63 only --
//   * collisions + neighbour candidates of the new state: every walker against the 64 slots of the wavefront's word (packed
//     fp32 circles, fp32 SAT filter, fp64 exact SAT: the stages of tile_collisions / walk_collisions, a quarter each);
//   * social force: the (walker, neighbour) pairs whose neighbour lies in the word, listed in LDS in (walker, neighbour)
//     order and evaluated 64 at a time over the wavefront's lanes (crowd_pair, no own / helper distinction); wavefront 0 then
//     adds every walker's terms word after word, i.e. in entity order: the reference's sum (social_force.py:64-84) bit for bit;
//   * the per-walker serial parts (goal update, move, statistics, stores, metrics, events) stay on wavefront 0; the rows of the
//     statics the walkers touch are rewritten by all 256 threads.
// Five workgroup barriers per step; ~2 k instructions on wavefront 0 and ~1.1 k on the others with ~30 walkers.  Everything a step
// decides is computed before anything of it is stored; a case this variant does not handle (see sgym_walk.hpp: equal
// geometries, operands outside crowd_pair's guards, more listed pairs than the LDS list holds) makes the workgroup stop at the
// last completed step and rollout_kernel_crowd finishes the chunk.  Bit-identical to rollout_kernel_crowd<4>
// (tests: test_crowd_walker_variant_is_invisible).
#pragma once
#include "sgym_walk.hpp"

namespace sg {

struct Walk4Lds {
    static constexpr int PCAP = 256; // listed pairs per wavefront (one 64-slot word of every walker's candidate row)
    // by entity slot
    float cx[WALK_SLOTS], cy[WALK_SLOTS]; // box centres (NaN: absent)
    float2 sc[WALK_SLOTS];                // sin, cos of the heading
    float2 half[WALK_SLOTS];              // half length, half width
    double px[WALK_SLOTS], py[WALK_SLOTS]; // reference point (px NaN: absent or not of type Pedestrian)
    unsigned long long wbits[WALK_SLOTS], wprev[WALK_SLOTS]; // statics: walker lanes that hit them (this / previous step)
    unsigned char lane_of[WALK_SLOTS];    // 255: the entity has no lane (a static, an absent entity, padding)
    // by walker lane
    unsigned char ent_of[64];
    double hd[64], ox[64], oy[64], sx[64], sy[64], ss[64]; // heading; crowd_pair's per-neighbour terms of a WALKER (statics: 0, ss0)
    double r2hi[64], r2lo[64], rad[64];   // radius rule of the lane's pedestrian
    unsigned long long rowbuf[4][64];     // word v of every walker's new collision row (written by wavefront v)
    unsigned short poff[4][64], pcnt[4][64]; // where wavefront v listed walker l's pairs, and how many
    alignas(16) double2 pres[4][PCAP];    // c1x, c1y of a listed pair
    uint32_t plist[4][PCAP];              // owner lane << 8 | neighbour bit; after evaluation: + flags (bit 16 inactive, 17 / 18 signs of c2)
    unsigned long long go_mask;           // walker lanes that take a step towards their goal
    double ss0;                           // the statics' step * step term of the coming step
    int cont;                             // the step loop goes on (decided by wavefront 0)
    int vote[2][4];
    double gon[128];                      // cos, sin of 2 pi i / 64 (p.gon)
};

// State.collisions() of the walker lanes against the 64 entity slots of word v, + the neighbour candidates of the coming step,
// from the LDS tables (the walkers' entries hold the state to test).  row / nbr: word v of the lane's rows.
__device__ __forceinline__ void walk4_collide_word(const Params &p, Walk4Lds &L, const WalkLane &W, int r, int v, int lane, bool scatter,
                                                   uint64_t &row, uint64_t &nbr, bool &bail)
{
    const int e = W.e;
    const bool act = e >= 0;
    const int es = act ? e : 0;
    const float fx = act ? L.cx[es] : __builtin_nanf(""), fy = act ? L.cy[es] : __builtin_nanf("");
    const float2 msc = L.sc[es];
    const float fs = msc.x, fc = msc.y;
    const bool pres = act && fx == fx; // (an absent walker published NaN centres)
    const float mag = __builtin_fabsf(fx) + __builtin_fabsf(fy);
    const float reach = W.rad_thr + 1.9073486e-6f * mag;
    const float thr = reach * reach;
    const float nreach = W.nbr_thr + 1.9073486e-6f * mag;
    const float nthr = nreach * nreach;
    row = 0;
    nbr = 0;
    // ---- broad phase: the 64 slots of the word, four at a time, packed fp32 (tile_collisions, all_pairs) ----
    uint64_t close;
    {
        const v2f fx2 = {fx, fx}, fy2 = {fy, fy}, thr2 = {thr, thr}, nthr2 = {nthr, nthr};
        const int base = v * 64;
        uint32_t out_w[2], nout_w[2]; // bit j = 1: slot j is OUTSIDE this lane's reach
#pragma unroll
        for (int w2 = 1; w2 >= 0; --w2) {
            uint32_t w = 0u, u = 0u;
#pragma unroll
            for (int q = 7; q >= 0; --q) {
                const int jb = base + w2 * 32 + q * 4;
                const v4f xs = *reinterpret_cast<const v4f *>(&L.cx[jb]);
                const v4f ys = *reinterpret_cast<const v4f *>(&L.cy[jb]);
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
                u = __builtin_amdgcn_alignbit(u, __float_as_uint(nb.y), 31);
                u = __builtin_amdgcn_alignbit(u, __float_as_uint(nb.x), 31);
                u = __builtin_amdgcn_alignbit(u, __float_as_uint(na.y), 31);
                u = __builtin_amdgcn_alignbit(u, __float_as_uint(na.x), 31);
            }
            out_w[w2] = w;
            nout_w[w2] = u;
        }
        uint64_t inside = ~(((uint64_t)out_w[1] << 32) | out_w[0]);
        uint64_t nin = ~(((uint64_t)nout_w[1] << 32) | nout_w[0]);
        if ((es >> 6) == v) { inside &= ~(1ull << (es & 63)); nin &= ~(1ull << (es & 63)); } // not with itself
        close = pres ? inside : 0;
        nbr = (pres && W.nbr_thr > 0.0f) ? nin : 0;
    }
    // ---- fp32 SAT filter (tile_collisions) ----
    uint64_t fuzzy = 0;
    if (sg_any(close != 0)) {
        uint64_t cand = close;
        while (sg_any(cand != 0)) {
            if (cand) {
                const int jl = __builtin_ctzll(cand);
                cand &= cand - 1;
                const int j = v * 64 + jl;
                const float2 oc = make_float2(L.cx[j], L.cy[j]), os = L.sc[j], oh = L.half[j];
                const float dx = oc.x - fx, dy = oc.y - fy;
                const float cd = __builtin_fabsf(fc * os.y + fs * os.x);
                const float sd = __builtin_fabsf(fs * os.y - fc * os.x);
                const float g0 = __builtin_fabsf(dx * fc + dy * fs) - (W.hl + oh.x * cd + oh.y * sd);
                const float g1 = __builtin_fabsf(dy * fc - dx * fs) - (W.hw + oh.x * sd + oh.y * cd);
                const float g2 = __builtin_fabsf(dx * os.y + dy * os.x) - (oh.x + W.hl * cd + W.hw * sd);
                const float g3 = __builtin_fabsf(dy * os.y - dx * os.x) - (oh.y + W.hl * sd + W.hw * cd);
                const float gap = __builtin_fmaxf(__builtin_fmaxf(g0, g1), __builtin_fmaxf(g2, g3));
                const float eps = 1e-3f + 1.9073486e-6f * (mag + __builtin_fabsf(oc.x) + __builtin_fabsf(oc.y)) + W.trig_eps;
                bool unsure = (gap <= eps) && (gap >= -eps);
                unsure = unsure || (dx == 0.0f && dy == 0.0f);
                if (unsure) fuzzy |= 1ull << jl;
                else if (gap < -eps) row |= 1ull << jl;
            }
        }
    }
    // ---- exact fp64 SAT on the pairs inside the margin; corners from the LDS poses ----
    bool eq = false;
    if (sg_any(fuzzy != 0)) {
        double A[8];
        {
            double s, c;
            sg_sincos(L.hd[lane], s, c);
            sg_corners(L.px[es] == L.px[es] ? L.px[es] : 0.0, L.py[es], s, c, W.bw, W.bl, W.bcx, W.bcy, A);
        }
        while (sg_any(fuzzy != 0)) {
            if (fuzzy) {
                const int jl = __builtin_ctzll(fuzzy);
                fuzzy &= fuzzy - 1;
                const int j = v * 64 + jl;
                const double *sb = p.stat + ((size_t)r * 4 + (j >> 6)) * (ST_COUNT * 64) + (j & 63);
                const int jlane = L.lane_of[j];
                double B[8], s, c;
                sg_sincos(jlane == 255 ? 0.0 : L.hd[jlane & 63], s, c);
                sg_corners(L.px[j], L.py[j], s, c, sb[ST_BW * 64], sb[ST_BL * 64], sb[ST_BCX * 64], sb[ST_BCY * 64], B);
                bool same = true;
#pragma unroll
                for (int k = 0; k < 8; ++k) same = same && (B[k] == A[k]);
                if (same) eq = true; // g == g_prime (utils.py:59) and the owner mapping: the full kernel's business
                else if (sg_quads_intersect(A, B)) row |= 1ull << jl;
            }
        }
    }
    if (sg_any(eq)) bail = true; // (made uniform by the caller's vote)
    // ---- the same hits from the statics' side ----
    if (scatter) {
        uint64_t m = row;
        while (m) {
            const int j = v * 64 + __builtin_ctzll(m);
            m &= m - 1;
            if (L.lane_of[j] == 255) atomicOr(&L.wbits[j], 1ull << lane);
        }
    }
}

// One workgroup of 256 threads per scenario of class 1 (<= 64 active entities): see the file header.
__device__ __forceinline__ void walk4_body(const Params &p, double timestep, int n_steps, int force, const WalkArgs &wa)
{
    using LDS = Walk4Lds;
    constexpr int PCAP = LDS::PCAP;
    __shared__ LDS L;
    const int r = blockIdx.x;
    if (wa.cls[r] != 1) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef SG_WALK4_ROTATE
    const int sw = r & 3; // the wavefront that carries the per-walker serial parts (experiment: spread them over the SIMDs)
#else
    const int sw = 0;
#endif
    const int step_target = wa.target[r];
    const int n_act = wa.n_active[r];
    const int e_raw = (int)wa.ent[(size_t)r * 128 + lane];
    const bool act = lane < n_act; // (every value 0..255 of the list is an entity: the count says where it ends)
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
    // ---- LDS tables of all 256 entities from the state in memory: thread = slot ----
    const uint64_t *smask = wa.smask + (size_t)r * 4;
    float rmax = 0.0f, omax = 0.0f, omax_ped = 0.0f; // largest bounding-circle radius / centre offset of the scenario (rollout_body)
    bool base_mine = false;
    {
        const int s = tid;
        const size_t b2 = (size_t)r * 4 + (s >> 6);
        const LanePtr st2(p.stat + b2 * (ST_COUNT * 64), (s & 63) * 8u);
        const LanePtr dy2(p.dyn + b2 * ((size_t)(SG_F_COLL + 4) * 64), (s & 63) * 8u);
        const int64_t m2 = fld<int64_t>(st2, ST_META);
        const bool stat = (smask[s >> 6] >> (s & 63)) & 1;
        const double bw = fld(st2, ST_BW), bl = fld(st2, ST_BL), bcx = fld(st2, ST_BCX), bcy = fld(st2, ST_BCY);
        rmax = (float)(0.5 * __builtin_sqrt(bl * bl + bw * bw)) * 1.000001f; // (padding slots: zero boxes, as in rollout_body's reductions)
        omax = (float)__builtin_sqrt(bcx * bcx + bcy * bcy) * 1.000001f;
        if (((m2 >> 8) & 0xff) == 1) omax_ped = omax;
        L.half[s] = make_float2((float)(0.5 * bl), (float)(0.5 * bw));
        L.lane_of[s] = 255;
        // a static: heading 0, velocity 0; everything else (absent, padding, active): filled / overwritten by its lane below
        float cx = __builtin_nanf(""), cy = cx;
        double px = __builtin_nan(""), py = 0.0;
        float fs = 0.0f, fc = 1.0f;
        uint64_t other = 0;
        if (stat) {
            px = fld(dy2, SG_F_POSE + 0);
            py = fld(dy2, SG_F_POSE + 1);
            sg_sincos_f32(0.0, fs, fc);
            cx = (float)px + ((float)bcx * fc - (float)bcy * fs);
            cy = (float)py + ((float)bcx * fs + (float)bcy * fc);
            // hits of non-statics that the row in memory still holds: rewritten by the first step (whatever it finds)
#pragma unroll
            for (int q = 0; q < 4; ++q) other |= fld<uint64_t>(dy2, SG_F_COLL + q) & ~smask[q];
        }
        L.sc[s] = make_float2(fs, fc);
        L.cx[s] = cx; L.cy[s] = cy;
        L.px[s] = px; L.py[s] = py;
        L.wbits[s] = 0;
        L.wprev[s] = other ? ~0ull : 0ull;
        const uint64_t *b = wa.base + ((size_t)r * WALK_SLOTS + s) * 4;
        base_mine = (b[0] | b[1] | b[2] | b[3]) != 0; // some static-static hit exists (terminal condition "collision")
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        rmax = __builtin_fmaxf(rmax, __shfl_xor(rmax, o, 64));
        omax = __builtin_fmaxf(omax, __shfl_xor(omax, o, 64));
        omax_ped = __builtin_fmaxf(omax_ped, __shfl_xor(omax_ped, o, 64));
    }
    {
        float *red = reinterpret_cast<float *>(L.pres[0]);
        const bool base_w = sg_any(base_mine);
        if (lane == 0) { red[wave] = rmax; red[8 + wave] = omax; red[16 + wave] = omax_ped; red[24 + wave] = base_w ? 1.0f : 0.0f; }
        if (tid < 128) L.gon[tid] = p.gon[tid];
        __syncthreads();
        float bany = 0.0f;
        for (int w = 0; w < 4; ++w) {
            rmax = __builtin_fmaxf(rmax, red[w]); omax = __builtin_fmaxf(omax, red[8 + w]);
            omax_ped = __builtin_fmaxf(omax_ped, red[16 + w]);
            bany = __builtin_fmaxf(bany, red[24 + w]);
        }
        base_mine = bany != 0.0f;
        __syncthreads();
    }
    const bool base_any = base_mine;
    {
        const float rad = (float)(0.5 * __builtin_sqrt(W.bl * W.bl + W.bw * W.bw)) * 1.000001f;
        const float off = (float)__builtin_sqrt(W.bcx * W.bcx + W.bcy * W.bcy) * 1.000001f;
        W.rad_thr = rad + rmax + 2e-3f + 2.0f * SG_TRIG32_ERR * (off + omax);
        W.trig_eps = SG_TRIG32_ERR * (12.0f * W.rad_thr + 4.0f * (off + omax));
        W.nbr_thr = is_ped ? (float)rad_c * 1.000001f + off + omax_ped + 2e-3f + 2.0f * SG_TRIG32_ERR * (off + omax_ped) : 0.0f;
        W.hl = (float)(0.5 * W.bl);
        W.hw = (float)(0.5 * W.bw);
    }
    if (wave == sw) {
        if (act) {
            L.lane_of[e] = (unsigned char)lane;
            L.ent_of[lane] = (unsigned char)e;
            const double r2 = rad_c * rad_c;
            L.r2hi[lane] = r2 * (1.0 + 1e-9);
            L.r2lo[lane] = r2 * 0.9975;
            L.rad[lane] = rad_c;
        } else {
            L.ent_of[lane] = 0;
            L.r2hi[lane] = 0.0; L.r2lo[lane] = 0.0; L.rad[lane] = 0.0;
        }
        L.hd[lane] = 0.0; L.ox[lane] = 0.0; L.oy[lane] = 0.0; L.sx[lane] = 0.0; L.sy[lane] = 0.0; L.ss[lane] = 0.0;
    }
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
    // ---- the walker lanes' state from memory (wavefront 0 owns it; rollout_body, continuing launch) ----
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
    uint64_t last_row[WALK_NW], row[WALK_NW];
#pragma unroll
    for (int w = 0; w < WALK_NW; ++w) { last_row[w] = sd.last_row[w]; row[w] = fld<uint64_t>(dy, SG_F_COLL + w); }
    int n_ev = sd.n_events, done = sd.done, steps = sd.n_steps;
    long long noise_pos = sd.noise_pos;
    sg_loads_done();

    // a walker's entries of the LDS tables for the state (npres, x, y, h, vx, vy), dtn = the dt of the step that will read them
    auto publish = [&](bool npres, double x, double y, double h, double vx, double vy, double dtn, bool &insane) {
        float fs, fc;
        sg_sincos_f32(h, fs, fc);
        const float bcxf = (float)W.bcx, bcyf = (float)W.bcy;
        const bool pres = act && npres;
        const float nanf_ = __builtin_nanf("");
        const float fx = pres ? (float)x + (bcxf * fc - bcyf * fs) : nanf_;
        const float fy = pres ? (float)y + (bcxf * fs + bcyf * fc) : nanf_;
        // the neighbour's terms of the repulsion, once per neighbour (tile_collisions)
        const double vmag = sg_norm2(vx, vy) + 0.0000000001;
        const double uox = vx / vmag, uoy = vy / vmag, stp = vmag * dtn;
        const double sxx = stp * uox, syy = stp * uoy;
        insane = pres & !(crowd_sane(x, 0x1p400) & crowd_sane(y, 0x1p400) & crowd_sane(sxx, 0x1p20) & crowd_sane(syy, 0x1p20) & (stp < 0x1p20));
        if (act) {
            L.cx[e] = fx;
            L.cy[e] = fy;
            L.sc[e] = make_float2(fs, fc);
            L.px[e] = (pres && W.is_ped_type) ? x : __builtin_nan("");
            L.py[e] = y;
            L.hd[lane] = h;
            L.ox[lane] = uox; L.oy[lane] = uoy;
            L.sx[lane] = sxx; L.sy[lane] = syy; L.ss[lane] = stp * stp;
        }
        if (lane == 0) { // the statics' step * step: |v| + 1e-10 with v = 0, times the coming step's dt (social_force.py:148-155)
            const double st0 = 0.0000000001 * dtn;
            L.ss0 = st0 * st0;
        }
    };

    bool bail = false;
    uint64_t nbr_v = 0; // word `wave` of the lane's neighbour candidate row (for the coming step)
    // ---- opening pass: LDS entries and neighbour candidates of the state in memory ----
    __syncthreads();
    {
        bool insane = false;
        if (wave == sw) publish(present, pose[0], pose[1], pose[3], velx, vely, (t + timestep) - t, insane);
        const bool iw = sg_any(insane);
        if (lane == 0) L.vote[0][wave] = iw ? 1 : 0;
        __syncthreads();
        bool cb = false;
        uint64_t tmp;
        walk4_collide_word(p, L, W, r, wave, lane, false, tmp, nbr_v, cb);
        const bool cw = sg_any(cb);
        if (lane == 0) L.vote[1][wave] = cw ? 1 : 0;
        __syncthreads();
        bail = (L.vote[0][0] | L.vote[0][1] | L.vote[0][2] | L.vote[0][3] | L.vote[1][0] | L.vote[1][1] | L.vote[1][2] | L.vote[1][3]) != 0;
    }
    const double *Kp = SG_TRIG;
    WalkTimers wt;
#ifdef SG_WALK_TIMERS
    for (int i = 0; i < 16; ++i) wt.acc[i] = 0;
    wt.last = __builtin_amdgcn_s_memtime();
#define W4T(i) do { if (wave < 2) WT(wave * 8 + (i)); } while (0)
#else
#define W4T(i) ((void)0)
#endif
    for (int k = 0;; ++k) {
        // ---- (A) wavefront 0: does the scenario step?  goal update + the force to the goal ----
        bool go = false;
        double fx = 0.0, fy = 0.0, vdes = 0.0;
        asm volatile("" : "+s"(Kp));
        ConstTbl K = (ConstTbl)Kp;
        const double next_t = t + timestep; // scenario_gym.py:229
        const double state_dt = t - prev_t;
        const double dt = next_t - t;
        if (wave == sw) {
            const bool run = k < n_steps && !bail && (force || !done) && steps < step_target; // (uniform: one scenario per workgroup)
            if (run && is_ped && present) {
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
            const uint64_t gm = __ballot(go);
            if (lane == 0) { L.go_mask = gm; L.cont = run ? 1 : 0; }
        }
        W4T(0);
        __syncthreads(); // B1
        W4T(1);
        if (!L.cont) break;
        // ---- (B) every wavefront: the pairs whose neighbour lies in its word, listed in (walker, neighbour) order ----
        bool pbail = false;
        {
            const uint64_t m = ((L.go_mask >> lane) & 1) ? nbr_v : 0;
            const int n = __builtin_popcountll(m);
            int scan = n;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int u = __shfl_up(scan, o, 64);
                if (lane >= o) scan += u;
            }
            const int total_all = __shfl(scan, 63, 64);
            pbail = total_all > PCAP; // (more pairs than the list holds: the full kernel's business)
            const int total = min(total_all, PCAP);
            const int off = scan - n;
            L.poff[wave][lane] = (unsigned short)min(off, PCAP);
            L.pcnt[wave][lane] = (unsigned short)max(min(n, PCAP - off), 0);
            uint64_t mm = m;
            for (int q = 0; sg_any(mm != 0); ++q) {
                if (mm) {
                    const int jb = __builtin_ctzll(mm);
                    mm &= mm - 1;
                    if (off + q < PCAP) L.plist[wave][off + q] = ((uint32_t)lane << 8) | (uint32_t)jb;
                }
            }
            tile_sync<1>();
            for (int idx = lane; idx - lane < total; idx += 64) { // (uniform trip count)
                const bool valid = idx < total;
                const uint32_t ent = L.plist[wave][min(idx, PCAP - 1)];
                const int ol = (int)((ent >> 8) & 63), j = wave * 64 + (int)(ent & 63);
                const int oe = L.ent_of[ol];
                const int jl = L.lane_of[j];
                const bool jstat = jl == 255;
                const int jq = jl & 63;
                const double rx = L.px[oe] - L.px[j], ry = L.py[oe] - L.py[j];
                const double nox = jstat ? 0.0 : L.ox[jq], noy = jstat ? 0.0 : L.oy[jq];
                const double nsx = jstat ? 0.0 : L.sx[jq], nsy = jstat ? 0.0 : L.sy[jq], nss = jstat ? L.ss0 : L.ss[jq];
                double c1x, c1y, c2x, c2y, d2;
                bool bad;
                crowd_pair(CC, rx, ry, nox, noy, nsx, nsy, nss, c1x, c1y, c2x, c2y, d2, bad);
                const bool outside = d2 > L.r2hi[ol], inside = d2 < L.r2lo[ol];
                const bool ring = valid & !(outside | inside);
                bool pact = valid & inside;
                if (sg_any(ring)) { // rare: between the inscribed circle and the vertices of the 64-gon Point.buffer(r)
                    if (ring) pact = sg_in_radius(L.px[oe], L.py[oe], L.rad[ol], L.px[j], L.py[j], L.gon);
                }
                pbail = pbail | (bad & pact); // (voted below)
                if (valid) {
                    L.pres[wave][idx] = make_double2(c1x, c1y);
                    L.plist[wave][idx] = ent | (pact ? 0u : 1u << 16) | (__builtin_signbit(c2x) ? 1u << 17 : 0u) |
                                         (__builtin_signbit(c2y) ? 1u << 18 : 0u);
                }
            }
            const bool pw = sg_any(pbail);
            if (lane == 0) L.vote[0][wave] = pw ? 1 : 0;
        }
        W4T(2);
        __syncthreads(); // B2
        W4T(3);
        // ---- (C) wavefront 0: the sums in entity order, noise, PedestrianController, statistics; the new state into the tables ----
        bool npres = false;
        double np_[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, fpx = 0.0, fpy = 0.0, ncspeed = cspeed;
        double d[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, vel[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        bool insane = false;
        long long noise_next = noise_pos; // (committed with the step)
        if (wave == sw) {
#pragma unroll 1
            for (int w = 0; w < 4; ++w) { // SocialForce._step :64-84 with sight weights: repulsion, then attraction, neighbour by neighbour
                const int off = L.poff[w][lane], cnt = go ? (int)L.pcnt[w][lane] : 0;
                for (int q = 0; sg_any(q < cnt); ++q) {
                    if (q < cnt) {
                        const uint32_t en = L.plist[w][off + q];
                        const double2 c1 = L.pres[w][off + q];
                        if (!(en & (1u << 16))) {
                            fx += c1.x; fy += c1.y;
                            fx += (en & (1u << 17)) ? -0.0 : 0.0; fy += (en & (1u << 18)) ? -0.0 : 0.0;
                        }
                    }
                }
            }
            // random fluctuations (rollout_body)
            PedNoise nz{0.0, 0.0, false};
            if (p.noise_mode == 1) {
                const uint64_t walk = __ballot(go);
                const int before = __builtin_popcountll(walk & ((1ull << lane) - 1)), count = __builtin_popcountll(walk);
                const long long at = noise_pos + 2 * before;
                if (go) {
                    const bool inside = at + 1 < p.noise_len;
                    const double *z = p.noise_normals + (size_t)r * (size_t)p.noise_len + (inside ? at : 0);
                    nz = PedNoise{p.noise_std_lon * (inside ? z[0] : 0.0), p.noise_std_lat * (inside ? z[1] : 0.0), true};
                }
                noise_next = noise_pos + 2 * count;
            } else if (p.noise_mode == 2) {
                double z0, z1;
                sg_noise_pair(p.noise_seed, (uint32_t)r, (uint32_t)es, (uint32_t)steps, z0, z1, K);
                nz = PedNoise{p.noise_std_lon * z0, p.noise_std_lat * z1, true};
            }
            // new poses: scenario_gym.py:233-245
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
            // State.update_poses / update_statistics, state.py:203-239
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
            publish(npres, np_[0], np_[1], np_[3], vel[0], vel[1], (next_t + timestep) - next_t, insane);
            const bool iw = sg_any(insane);
            if (lane == 0) L.vote[0][sw] |= iw ? 1 : 0; // (joins the pair vote of this wavefront)
        }
        W4T(4);
        __syncthreads(); // B3
        W4T(1);
        // ---- (D) every wavefront: State.collisions of the new state against its word; nothing of the step is stored yet ----
        uint64_t nrow_v = 0, nnbr_v = 0;
        bool cbail = false;
        if (!(L.vote[0][0] | L.vote[0][1] | L.vote[0][2] | L.vote[0][3])) // (uniform; else the step is abandoned below)
            walk4_collide_word(p, L, W, r, wave, lane, true, nrow_v, nnbr_v, cbail);
        L.rowbuf[wave][lane] = nrow_v;
        {
            const bool cw = sg_any(cbail);
            if (lane == 0) L.vote[1][wave] = cw ? 1 : 0;
        }
        W4T(5);
        __syncthreads(); // B4
        W4T(1);
        if ((L.vote[0][0] | L.vote[0][1] | L.vote[0][2] | L.vote[0][3] | L.vote[1][0] | L.vote[1][1] | L.vote[1][2] | L.vote[1][3]) != 0) {
            bail = true; // (uniform) the step is not committed: rollout_kernel_crowd redoes it from the state in memory
            break;
        }
        nbr_v = nnbr_v;
        // ---- (E) wavefront 0: commit, stores, metrics, terminal conditions, events ----
        if (wave == sw) {
            present = npres;
            cspeed = ncspeed;
            noise_pos = noise_next;
            if (npres) {
#pragma unroll
                for (int c = 0; c < 6; ++c) pose[c] = np_[c];
                dist += sg_norm3(d[0], d[1], d[2]);
                velx = vel[0];
                vely = vel[1];
            }
#pragma unroll
            for (int w = 0; w < WALK_NW; ++w) row[w] = L.rowbuf[w][lane];
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
        }
        prev_t = t;
        t = next_t;
        ++steps;
        // ---- the rows of the statics the walkers touch (or touched in the previous step): thread = slot ----
        {
            const int s = tid;
            const uint64_t wnow = L.wbits[s];
            if ((wnow | L.wprev[s]) != 0 && L.lane_of[s] == 255) {
                const uint64_t *b = wa.base + ((size_t)r * WALK_SLOTS + s) * 4;
                uint64_t rw[4] = {b[0], b[1], b[2], b[3]};
                uint64_t m = wnow;
                while (m) {
                    const int l = __builtin_ctzll(m);
                    m &= m - 1;
                    const int o = L.ent_of[l];
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if ((o >> 6) == v) rw[v] |= 1ull << (o & 63);
                }
                L.wprev[s] = wnow;
                L.wbits[s] = 0;
                double *drow = p.dyn + ((size_t)r * 4 + (s >> 6)) * ((size_t)(SG_F_COLL + 4) * 64) + (s & 63);
#pragma unroll
                for (int v = 0; v < 4; ++v) reinterpret_cast<uint64_t *>(drow)[(SG_F_COLL + v) * 64] = rw[v];
            }
        }
        if (wave == sw) {
            // ego metrics, scenario_gym.py:251-252
            if (is_ego && present) {
                const double speed = sg_norm3(vel[0], vel[1], vel[2]);
                const double w = m_t / t; // EgoAvgSpeed._step, metrics/trajectory.py:19-24
                m_avg += (1.0 - w) * (speed - m_avg);
                m_t = t;
                m_max = __builtin_fmax(speed, m_max);
            }
            // check_terminal, state.py:268-270, 397-408
            int ndone = 0;
            if ((p.term_mask & SG_TERM_MAX_LENGTH) && (t + dt > length)) ndone = 1;
            if (p.term_mask & (SG_TERM_COLLISION | SG_TERM_EGO_COLLISION)) {
                const bool any_mine = act && (row[0] | row[1] | row[2] | row[3]) != 0;
                const bool a0 = sg_any(any_mine), a1 = sg_any(act && e == 0 && present && any_mine);
                if ((p.term_mask & SG_TERM_COLLISION) && (a0 || base_any)) ndone = 1;
                if ((p.term_mask & SG_TERM_EGO_COLLISION) && a1) ndone = 1;
            }
            done = ndone;
            // CollisionMetric._step, metrics/collision.py:70-75 (ego lane)
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
        }
        W4T(6);
        // (the next round's barrier B1 separates the statics pass above from the next scatter)
    }
#ifdef SG_WALK_TIMERS
    if (lane == 0 && wave < 2 && wa.stats64) for (int i = 0; i < 16; ++i) if (wt.acc[i]) atomicAdd(wa.stats64 + i, wt.acc[i]);
    if (tid == 0 && wa.stats64) atomicAdd(wa.stats64 + 7, (unsigned long long)(steps - (step_target - n_steps > 0 ? step_target - n_steps : 0))); // (steps of this launch, roughly)
#endif
    // ---- write back what lives in registers (wavefront 0) ----
    if (wave == sw && act) {
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
__global__ __launch_bounds__(256, 2) void walk4_kernel(Params p, double timestep, int n_steps, int force, WalkArgs wa)
{
    walk4_body(p, timestep, n_steps, force, wa);
}
#endif

} // namespace sg
