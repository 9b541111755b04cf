// sgym_crowd.hpp -- Workgroup votes, the neighbour loops of the social force (serial, balanced, crowd_pair / crowd_pairs), ped_force, ped_move.
// Part of the gfx950 device code of the batched rollout engine; included by sgym_device.hpp (in order: every part builds on
// the ones before it), never on its own.
#pragma once

namespace sg {

// Barrier between the lanes of one tile's workgroup.  A single wavefront (WV == 1) needs no s_barrier and no
// s_waitcnt: the LDS executes the instructions of one wavefront in issue order, so a ds_read issued after another
// lane's ds_write / ds_or already sees it; compiler barriers keep the compiler from reordering them.
template <int WV>
__device__ __forceinline__ void tile_sync()
{
    if (WV == 1) {
        // (compiler barriers only: a wavefront-scope fence made LLVM wait for lgkmcnt(0) here -- a full LDS round trip the
        // hardware does not need, its LDS instructions execute in issue order)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
    } else {
        __syncthreads();
    }
}

// Inclusive prefix sum over the 64 lanes of a wavefront with six DPP adds (row_shr 1, 2, 4, 8 inside each row of 16 lanes, then
// row_bcast 15 / 31 across the rows): no LDS round trips and no per-distance lane indices to keep in registers, which is what
// a __shfl_up loop costs.  The wavefront's total is lane 63's value.
__device__ __forceinline__ int wave_scan_incl(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false); // row_shr:1 (lanes without a source add 0)
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false); // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false); // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false); // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
    return x;
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
    const bool act = valid & (L.isped[j] != 0) & sg_in_radius(ipx, ipy, irad, ox, oy, LDS::GON ? L.gon : p.gon);
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

#ifndef SG_CROWD_ILP
#define SG_CROWD_ILP 1 // (pedestrian, neighbour) pairs a lane evaluates side by side.  fp64 issue, not latency, bounds the chain
                       // (tools/dbg/pair_bench.hip): 1 -> 4.38, 2 -> 4.35, 3 -> 4.17 G on 1024 x 256, three alternating repetitions
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
    const int total = __builtin_amdgcn_readlane(wave_scan_incl(n), 63);
    if (total == 0) return; // wave-uniform
    const int T = (total + 63) >> 6;
    const int excess = max(n - T, 0), spare = max(T - n, 0);
    const int scan = wave_scan_incl(excess | (spare << 16)); // both prefix sums at once (each < 2^15)
    const int listed_all = min(__builtin_amdgcn_readlane(scan, 63) & 0xffff, CAP);
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
                                          L.px[jj[u]], L.py[jj[u]], LDS::GON ? L.gon : p.gon);
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
        double w1x = 0.0, w1y = 0.0;
        // (the crowd variants: the route's waypoints loaded once per step -- one memory latency instead of three)
        const bool two = CROWD && nwp == 2 && goal_idx <= 1, few = CROWD && nwp > 2 && nwp <= 4 && goal_idx <= nwp - 1;
        if (two) goal_idx = ped_goal_update2(wp, pose[0], pose[1], w1x, w1y);
        else if (few) goal_idx = ped_goal_update_reg<4>(wp, nwp, pose[0], pose[1], w1x, w1y);
        else if (goal_idx <= nwp - 1) goal_idx = ped_goal_update(wp, nwp, pose[0], pose[1]);
        if (goal_idx <= nwp - 1) {
            go = true;
            const bool have = two | few;
            double gx = (have ? w1x : wp[2 * goal_idx]) - pose[0], gy = (have ? w1y : wp[2 * goal_idx + 1]) - pose[1]; // _force_to_goal, :119-138
            double gn = sg_norm2(gx, gy);
            if (gn == 0) gn += 0.000000001;
            vdes = L.ctrl[SG_C_PED_SPEED_DESIRED - SG_C_PED_SPEED_DESIRED][sl];
            const double inv_tau = 1 / sf.relaxation_time;
            fx = inv_tau * (vdes * (gx / gn) - velx);
            fy = inv_tau * (vdes * (gy / gn) - vely);
            if (!CROWD && p.ped_behaviour == SG_PED_RANDOM_WALK) { // RandomWalk._step: the vector to the goal point is all it
                fx = gx;                                           // looks at (random_walk.py:40-41; ped_move takes its angle)
                fy = gy;
            }
            if (!CROWD) sg_sincos(L.ctrl[SG_C_PED_HEAD_ROT - SG_C_PED_SPEED_DESIRED][sl], hs, hc, K);
            radius = L.ctrl[SG_C_PED_RADIUS - SG_C_PED_SPEED_DESIRED][sl];
        }
    }
    if (!CROWD && p.ped_behaviour == SG_PED_RANDOM_WALK) return; // (launch-uniform: no neighbours, no boundary terms)
    const double k2_scale = sf.ped_repulse_V / sf.ped_repulse_sigma;
    if (CROWD && crowd_fast) { // wave-uniform: the guards of crowd_pair hold
        PH(0);
        crowd_pairs<WV>(p, L, CC, sl, nbr, go, k2_scale, pose[0], pose[1], fx, fy);
        PH(6);
    } else {
        if (CROWD && go) sg_sincos(L.ctrl[SG_C_PED_HEAD_ROT - SG_C_PED_SPEED_DESIRED][sl], hs, hc, K);
        // the shortcuts of ped_pair need the sight-weight branch (c2 = w2 * att) and hold for the whole wavefront
        const bool plain = sg_all(hs == 0.0 && hc == 1.0) && sf.ped_attract_C == 0.0 && sf.sight_weight > 0.0 &&
                           sf.sight_weight_use != 0.0;
        if (!CROWD && plain && !p.ped_serial)
            ped_pairs_balanced<WV>(p, L, sl, tile0, nbr, go, k2_scale, pose[0], pose[1], radius, fx, fy);
        else // (CROWD: the guards of crowd_pair do not hold, or SG_PED_SERIAL: the plain serial loop)
            ped_pairs_serial<WV>(p, L, tile0, nbr, go, plain, k2_scale, pose[0], pose[1], radius, hs, hc, fx, fy);
    }
    if (p.road && go) { // (launch-uniform: the batch has road networks) after the neighbours, social_force.py:83-104
        if (LDS::ROAD_TAB) ped_boundary_terms<true>(p, r, pose[0], pose[1], fx, fy, L.road_tab, L.road_info, &L.road_m, &L.road_net);
        else ped_boundary_terms(p, r, pose[0], pose[1], fx, fy);
    }
}

// The random fluctuations of one pedestrian's step: scale * z of np.random.normal(loc, scale) = loc + scale * z for the
// speed (s) and the heading (h); on == false: no generator in use (std 0), the locations alone.
struct PedNoise {
    double s, h;
    bool on;
};

// PedestrianAgent.step, part 2 (one lane): speed and heading from the force (:110-114, or zero at the goal,
// agent.py:65-68) + PedestrianController._step (pedestrian/controller.py:25-46).
// SocialForce: speed_rand = np.random.normal(bias_lon, std_lon) joins |F|, heading_rand = np.random.normal(bias_lat, std_lat)
// the force's angle (social_force.py:106-113).  RandomWalk (RW: compiled in for the variants that can run it; fx, fy = the
// vector to the goal point): the locations carry the signal, speed = np.random.normal(speed_desired + bias_lon, std_lon),
// heading = np.random.normal(angle + bias_lat, std_lat), no max_speed_factor, agent.force untouched (random_walk.py:37-43).
// the part of a behaviour model ped_move reads: per LANE where a batch mixes models (sg_set_ped_models)
struct PedMoveModel {
    int behaviour;
    double bias_lon, bias_lat, max_speed_factor;
};
__device__ __forceinline__ PedMoveModel ped_move_model(const Params &p)
{
    return PedMoveModel{p.ped_behaviour, p.sf.bias_lon, p.sf.bias_lat, p.sf.max_speed_factor};
}

template <bool RW = true>
__device__ __forceinline__ void ped_move(const PedMoveModel &pm, bool go, double fx, double fy, double vdes, double maxs,
                                         const double *pose, double state_dt, double &cspeed, double &fxo, double &fyo,
                                         double *np_, ConstTbl K, PedNoise nz)
{
    double speed = 0.0, heading = 0.0;
    fxo = fyo = 0.0;
    if (RW && pm.behaviour == SG_PED_RANDOM_WALK) {
        if (go) {
            const double loc_s = vdes + pm.bias_lon, loc_h = sg_atan2(fy, fx, K) + pm.bias_lat;
            speed = nz.on ? loc_s + nz.s : loc_s;
            heading = nz.on ? loc_h + nz.h : loc_h;
        }
    } else if (go) {
        const double speed_rand = nz.on ? pm.bias_lon + nz.s : pm.bias_lon;
        const double heading_rand = nz.on ? pm.bias_lat + nz.h : pm.bias_lat;
        speed = __builtin_fmin(sg_norm2(fx, fy) + speed_rand, vdes * pm.max_speed_factor);
        heading = sg_atan2(fy, fx, K) + heading_rand;
        fxo = fx;
        fyo = fy;
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

template <bool RW = true>
__device__ __forceinline__ void ped_move(const Params &p, bool go, double fx, double fy, double vdes, double maxs,
                                         const double *pose, double state_dt, double &cspeed, double &fxo, double &fyo,
                                         double *np_, ConstTbl K, PedNoise nz)
{
    ped_move<RW>(ped_move_model(p), go, fx, fy, vdes, maxs, pose, state_dt, cspeed, fxo, fyo, np_, K, nz);
}

} // namespace sg
