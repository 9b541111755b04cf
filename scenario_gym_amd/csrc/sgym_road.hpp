// sgym_road.hpp -- Road surfaces: exact point-in-union tests on the cell grid, the boundary terms of the social force.
// Part of the gfx950 device code of the batched rollout engine; included by sgym_device.hpp (in order: every part builds on
// the ones before it), never on its own.
#pragma once

namespace sg {

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

// the layers of `want` whose union strictly contains the point (one thread), in two halves: the loads that depend on the
// point's cell alone (rn_probe: the cell word and the range of its candidates, three gathers in flight together), and
// what follows from them (rn_resolve).  A caller with work of its own puts it in between (ped_boundary_terms).
struct RoadProbe { bool ok; int ix, iy; uint32_t m, k0, k1; };
__device__ __forceinline__ RoadProbe rn_probe(const RoadIndex &R, const RoadNet &N, double px, double py)
{
    RoadProbe q{false, 0, 0, 0u, 0u, 0u};
    if (!rn_cell_of(N, px, py, q.ix, q.iy)) return q;
    const int64_t cell = N.cell_base + (int64_t)q.iy * N.nx + q.ix;
    q.ok = true;
    q.m = R.cells[cell];
    q.k0 = R.cell_off[cell];
    q.k1 = R.cell_off[cell + 1];
    return q;
}
__device__ inline uint32_t rn_resolve(const RoadIndex &R, const RoadNet &N, const RoadProbe &q, uint32_t want, double px, double py)
{
    if (!q.ok) return 0u;
    const int ix = q.ix, iy = q.iy;
    uint32_t in = q.m & 0xffu & want, todo = (q.m >> 8) & want & ~in;
    if (todo) {
        for (uint32_t k = q.k0; k < q.k1 && todo; ++k) {
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
__device__ inline uint32_t rn_layers_at(const RoadIndex &R, int net, uint32_t want, double px, double py)
{
    if (net < 0) return 0u;
    const RoadNet N = R.nets[net];
    int ix, iy;
    if (!rn_cell_of(N, px, py, ix, iy)) return 0u;
    const int64_t cell = N.cell_base + (int64_t)iy * N.nx + ix;
    const uint32_t m = R.cells[cell];
    RoadProbe q{true, ix, iy, m, 0u, 0u};
    if ((m >> 8) & want & ~(m & 0xffu & want)) { q.k0 = R.cell_off[cell]; q.k1 = R.cell_off[cell + 1]; }
    return rn_resolve(R, N, q, want, px, py);
}

// One ring edge (a, b) of GEOS DistanceOp's nearest-point search folded into (best, cx, cy): Distance::pointToSegment, nearest
// first on ties, LineSegment::closestPoint -- the operation sequence of the oracle (sgo_boundary_terms).
__device__ __forceinline__ void nearest_edge_exact(double ax, double ay, double bx, double by, double px, double py, double &best,
                                                   double &cx, double &cy)
{
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

// The boundary terms of SocialForce._step (pedestrian/social_force.py:86-104, _force_boundary :190-211) for one
// pedestrian at (px, py) of scenario r.  nearest_points(surface, point) is GEOS DistanceOp: a point inside (or on) an
// areal geometry is its own nearest point, so the walkable term -- evaluated only INSIDE the walkable surface -- is the
// zero vector (+0.0 is still added, as the reference does), and so is the impenetrable term inside a building (-0.0);
// outside, every ring edge of the buildings in order: Distance::pointToSegment, nearest first on ties,
// LineSegment::closestPoint.  Same operation sequence as the oracle.
//
// The search visits the edges in order and keeps the FIRST one that attains the smallest rounded distance; only that edge
// decides the result (cx, cy are overwritten, never accumulated).  The reference's sequence costs two IEEE divisions and up
// to five square roots per edge, in branches a crowd's lanes take all of (~160 instructions per edge and wavefront: on the
// c5roads batch -- 16 edges -- the search was 43 % of the step).  So a FILTER first (networks of up to 64 such edges): the
// squared distance to every edge in plain fp64 with the edge's precomputed direction and reciprocal length (11 operations,
// no division, no root), twice -- the smallest, then who is within `margin` of it.  margin = 1e-11 x a bound on every
// squared length in play: the filter's own error is < 4e-15 of that bound and the reference's rounding < 2e-15 of it, so an
// edge outside the margin has a rounded distance strictly above the minimum and cannot be the one kept, whatever the
// order.  The reference's sequence then runs over the candidates only, in edge order: the point's own edge, both edges of a
// corner, the walls on either side of a street when they are equally far.  NaN anywhere makes everything a candidate.
// `tab` / `info` / `tab_m` / `tab_net` (rollout_kernel_crowd / _models, one scenario per workgroup): the scenario's building edges
// and its network's grid header staged in LDS once per launch (TileLds::road_tab / road_info / road_m / road_net, filled by rollout_body_l) -- as loads from device memory every
// edge of the two filter passes waited a full memory latency (the compiler cannot make them scalar loads: the kernel stores
// to global memory): c5roads 4.72 -> 5.0 G.
__device__ __forceinline__ double nearest_edge_approx_d2(double ax, double ay, double dx, double dy, double inv, double px, double py)
{
    const double dxp = px - ax, dyp = py - ay;
    const double t = __builtin_fmin(__builtin_fmax(__builtin_fma(dxp, dx, dyp * dy) * inv, 0.0), 1.0);
    const double qx = __builtin_fma(-t, dx, dxp), qy = __builtin_fma(-t, dy, dyp);
    return __builtin_fma(qx, qx, qy * qy);
}
// STAGED: a caller that stages (its scenarios without a staged table -- more than 64 edges -- walk every edge: one copy less
// of the filter in kernels whose registers are full).
template <bool STAGED = false>
__device__ inline void ped_boundary_terms(const Params &p, int r, double px, double py, double &fx, double &fy,
                                          const double *tab = nullptr, const int *info = nullptr, const double *tab_m = nullptr,
                                          const RoadNet *tab_net = nullptr)
{
    if (!p.road) return;
    const RoadIndex RI = *p.road;
    int net, n_tab = -1;
    uint32_t flags;
    if (info) { // (the same values in every lane: scalars)
        n_tab = __builtin_amdgcn_readfirstlane(info[0]);
        net = __builtin_amdgcn_readfirstlane(info[1]);
        flags = (uint32_t)__builtin_amdgcn_readfirstlane(info[2]);
        if (net < 0) return;
    } else {
        net = RI.net_of_scen[r];
        if (net < 0) return;
        flags = RI.net_flags[net];
    }
    if (!flags) return;
    const auto margin_of = [&](double m) {
        const double sxb = __builtin_fabs(px) + m, syb = __builtin_fabs(py) + m;
        return 1e-11 * (sxb * sxb + syb * syb + 8.0 * (m * m));
    };
    uint32_t in;
    uint64_t cand = 0;
    if (n_tab >= 0) { // the staged table: [k][5] = ax, ay, bx, by, ~1 / |b - a|^2
        // (the cell lookup's loads first, the filter passes while they are in flight, what follows from them after)
        const RoadNet N = *tab_net;
        const RoadProbe q = rn_probe(RI, N, px, py);
        const double margin = margin_of(*tab_m);
        double dmin = __builtin_inf();
#pragma unroll 4
        for (int k = 0; k < n_tab; ++k) {
            const double *e = tab + k * 5;
            dmin = __builtin_fmin(dmin, nearest_edge_approx_d2(e[0], e[1], e[2] - e[0], e[3] - e[1], e[4], px, py));
        }
        const double thr = dmin + margin;
#pragma unroll 4
        for (int k = 0; k < n_tab; ++k) {
            const double *e = tab + k * 5;
            if (!(nearest_edge_approx_d2(e[0], e[1], e[2] - e[0], e[3] - e[1], e[4], px, py) > thr)) cand |= 1ull << k;
        }
        in = rn_resolve(RI, N, q, SG_LAYER_WALKABLE | SG_LAYER_IMPENETRABLE, px, py);
    } else {
        in = rn_layers_at(RI, net, SG_LAYER_WALKABLE | SG_LAYER_IMPENETRABLE, px, py);
    }
    if ((flags & 1u) && (in & SG_LAYER_WALKABLE)) { fx += 0.0; fy += 0.0; }
    if (!(flags & 2u)) return;
    if (in & SG_LAYER_IMPENETRABLE) { fx += -0.0; fy += -0.0; return; }
    double best = __builtin_inf(), cx = px, cy = py;
    if (n_tab >= 0) {
        while (cand) {
            const double *e = tab + __builtin_ctzll(cand) * 5;
            cand &= cand - 1;
            nearest_edge_exact(e[0], e[1], e[2], e[3], px, py, best, cx, cy);
        }
    } else {
        const int64_t e0 = RI.imp_off[net], e1 = RI.imp_off[net + 1];
        if (!STAGED && e1 - e0 <= 64) {
            const double margin = margin_of(RI.imp_m[net]);
            double dmin = __builtin_inf();
            for (int64_t i = e0; i < e1; ++i) {
                const double *e = RI.imp_edges + i * 4, *a = RI.imp_aux + i * 4; // a: bx - ax, by - ay, ~1 / |b - a|^2
                dmin = __builtin_fmin(dmin, nearest_edge_approx_d2(e[0], e[1], a[0], a[1], a[2], px, py));
            }
            const double thr = dmin + margin;
            for (int64_t i = e0; i < e1; ++i) {
                const double *e = RI.imp_edges + i * 4, *a = RI.imp_aux + i * 4;
                if (!(nearest_edge_approx_d2(e[0], e[1], a[0], a[1], a[2], px, py) > thr)) cand |= 1ull << (int)(i - e0);
            }
            while (cand) {
                const double *e = RI.imp_edges + (e0 + __builtin_ctzll(cand)) * 4;
                cand &= cand - 1;
                nearest_edge_exact(e[0], e[1], e[2], e[3], px, py, best, cx, cy);
            }
        } else {
            for (int64_t i = e0; i < e1; ++i) {
                const double *e = RI.imp_edges + i * 4;
                nearest_edge_exact(e[0], e[1], e[2], e[3], px, py, best, cx, cy);
            }
        }
    }
    const double rx = px - cx, ry = py - cy, rn = sg_norm2(rx, ry);
    const double ux = rx / (rn + 0.0000000001), uy = ry / (rn + 0.0000000001);
    const double k = p.sf.imp_boundary_repulse_U / p.sf.imp_boundary_repulse_R, ex = sg_exp(-rn / p.sf.imp_boundary_repulse_R);
    fx += 1.0 * (k * ux * ex);
    fy += 1.0 * (k * uy * ex);
}

} // namespace sg
