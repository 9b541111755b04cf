// sgym_collide.hpp -- State.collisions() for one tile: broad phase (stripe masks / all-pairs walk), fp32 SAT filter, fp64 exact path, owner mapping.
// Part of the gfx950 device code of the batched rollout engine; included by sgym_device.hpp (in order: every part builds on
// the ones before it), never on its own.
#pragma once

namespace sg {

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
    // (vehicle tiles of up to 16 lanes always walk: four groups of four slots, no atomics, no mask reads, no candidate loop --
    // 256 x 16 replay 3.10 -> 3.31 G, 16384 x 16 70.6 -> 73.1 G; from 32 lanes on the stripe masks win, HISTORY.md round 4)
    const bool all_pairs = odd || (PED && dense) || (!PED && TS <= 16);
    if (!all_pairs) { // block_any / the barrier below also publish the LDS writes above
        // ---- stripe masks: O(tile) instead of O(tile^2) ----
        if (WV == 1) tile_sync<WV>();
        const int wsl = (WV == 1) ? 0 : (slot >> 6);              // word of this slot inside the tile's row
        const uint64_t mybit = 1ull << ((WV == 1) ? (sl & 63) : (slot & 63));
        if (present) { // (a tile of one wavefront: wavefront scope -- no wait behind the two ds_or)
            if (WV == 1) {
                __hip_atomic_fetch_or(&L.xtab[ix & 63][wsl], mybit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_fetch_or(&L.ytab[iy & 63][wsl], mybit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            } else {
                atomicOr(&L.xtab[ix & 63][wsl], mybit);
                atomicOr(&L.ytab[iy & 63][wsl], mybit);
            }
        }
        PH(2); tile_sync<WV>(); PH(13);
#pragma unroll
        for (int w = 0; w < WV; ++w) {
            // (all six words in flight before the first is used: one wait instead of two)
            const uint64_t x0 = L.xtab[(ix - 1) & 63][w], x1 = L.xtab[ix & 63][w], x2 = L.xtab[(ix + 1) & 63][w];
            const uint64_t y0 = L.ytab[(iy - 1) & 63][w], y1 = L.ytab[iy & 63][w], y2 = L.ytab[(iy + 1) & 63][w];
            uint64_t m = (x0 | x1 | x2) & (y0 | y1 | y2);
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
        // (the walk costs ~5 instructions per slot and no barrier; a candidate ~25 and the stripe masks two workgroup barriers.
        // Measured on 1024 x 256 (tools/dbg/dense_sweep.sh): switching at TS/10 ... TS/40 candidates and back below TS/24 ...
        // TS/256 neighbours all give 4.35 G, the earlier 2 TS/5 and TS/12 3.98 G, always walking 4.25 G)
        if (PED) dense = iters > TS / 16; // (voted by the next call)
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
    if (PED) { // back to the stripe masks once nobody has more than TS/64 neighbour candidates (hysteresis)
        int cnt = 0;
#pragma unroll
        for (int w = 0; w < WV; ++w) cnt += __builtin_popcountll(nbr_out[w]);
        dense = cnt > TS / 64; // (a wish: voted by the next call)
        PH(10);
    }
    }
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
    const float eps_own = __builtin_fmaf(1.9073486e-6f, mag, 1e-3f + trig_eps); // this lane's share of the margin (the candidate adds its own)
    if (sg_any(any_cand)) {
#pragma unroll
        for (int w = 0; w < WV; ++w) {
            while (sg_any(cand[w] != 0)) {
                if (cand[w]) {
                    const int jl = __builtin_ctzll(cand[w]);
                    cand[w] &= cand[w] - 1;
                    const int j = tile0 + w * 64 + jl;
                    const float2 oc = L.cen[j], os = L.sc[j], oh = L.half[j];
                    float dx = oc.x - fx, dy = oc.y - fy;
                    // (explicit fused multiply-adds: the filter only has to be conservative inside `eps`, which dwarfs an fp32
                    // rounding either way -- 38 instead of 54 instructions per candidate; whatever it leaves undecided the fp64
                    // test decides, so no output depends on this arithmetic)
                    float cd = __builtin_fabsf(__builtin_fmaf(fc, os.y, fs * os.x));   // |cos(delta heading)|
                    float sd = __builtin_fabsf(__builtin_fmaf(fs, os.y, -(fc * os.x))); // |sin(delta heading)|
                    float g0 = __builtin_fabsf(__builtin_fmaf(dx, fc, dy * fs)) - __builtin_fmaf(oh.y, sd, __builtin_fmaf(oh.x, cd, hl));
                    float g1 = __builtin_fabsf(__builtin_fmaf(dy, fc, -(dx * fs))) - __builtin_fmaf(oh.y, cd, __builtin_fmaf(oh.x, sd, hw));
                    float g2 = __builtin_fabsf(__builtin_fmaf(dx, os.y, dy * os.x)) - __builtin_fmaf(hw, sd, __builtin_fmaf(hl, cd, oh.x));
                    float g3 = __builtin_fabsf(__builtin_fmaf(dy, os.y, -(dx * os.x))) - __builtin_fmaf(hw, cd, __builtin_fmaf(hl, sd, oh.y));
                    float gap = __builtin_fmaxf(__builtin_fmaxf(g0, g1), __builtin_fmaxf(g2, g3));
                    // fp32 rounding of the centres + trig_eps: the hardware sin/cos error on every product
                    float eps = __builtin_fmaf(1.9073486e-6f, __builtin_fabsf(oc.x) + __builtin_fabsf(oc.y), eps_own);
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

} // namespace sg
