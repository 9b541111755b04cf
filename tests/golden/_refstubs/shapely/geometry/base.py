"""Geometry containers + exact-rational predicates for convex rings."""
import math
from fractions import Fraction

import numpy as np


def _F(v):
    return Fraction(float(v))


class _Coords(list):
    @property
    def xy(self):
        return ([c[0] for c in self], [c[1] for c in self])


class BaseGeometry:
    is_valid = True
    is_empty = False

    def buffer(self, *a, **k):
        return self

    def simplify(self, *a, **k):
        return self

    def __ne__(self, other):
        return not self.__eq__(other)


class LineString(BaseGeometry):
    def __init__(self, coords=()):
        if isinstance(coords, LineString):
            coords = coords.coords
        self.coords = _Coords(
            tuple(float(v) for v in np.asarray(c).ravel()[:2]) for c in coords
        )

    @property
    def xy(self):
        return self.coords.xy

    @property
    def length(self):
        c = np.array(self.coords)
        return float(np.linalg.norm(np.diff(c, axis=0), axis=1).sum()) if len(c) > 1 else 0.0

    def interpolate(self, s, normalized=False):
        """Point at arc length s along the polyline (clamped to its ends)."""
        c = np.array(self.coords)
        seg = np.diff(c, axis=0)
        ln = np.sqrt((seg ** 2).sum(axis=1))
        if normalized:
            s = s * float(ln.sum())
        if s <= 0.0 or len(c) == 1:
            return Point(c[0])
        acc = 0.0
        for k, L in enumerate(ln):
            if s < acc + L or (k == len(ln) - 1 and s <= acc + L):
                u = 0.0 if L == 0.0 else (s - acc) / L
                return Point(c[k] + u * seg[k])
            acc += L
        return Point(c[-1])

    def project(self, pt):
        """Arclength of the nearest point of the polyline (analytic, fp64)."""
        p = np.array([pt.x, pt.y])
        c = np.array(self.coords)
        best, best_s, acc = None, 0.0, 0.0
        for a, b in zip(c[:-1], c[1:]):
            d = b - a
            # GEOS arithmetic (LineSegment::projectionFactor, Coordinate::distance): plain products and sums.  numpy's
            # `@` on 2-vectors is a BLAS dot (an fma chain on this box) and can differ by one ulp -- enough to make the
            # projection of a point beyond the end of the line come out one ulp SHORT of the line's own length, which
            # PedestrianAgent compares with np.linalg.norm-based arc lengths (pedestrian/agent.py:59-62).
            L2 = float(d[0] * d[0] + d[1] * d[1])
            u = 0.0 if L2 == 0 else min(1.0, max(0.0, float((p[0] - a[0]) * d[0] + (p[1] - a[1]) * d[1]) / L2))
            q = a + u * d
            dist = float(np.hypot(*(p - q)))
            if best is None or dist < best:
                best, best_s = dist, acc + u * math.sqrt(L2)
            acc += math.sqrt(L2)
        return best_s

    def __eq__(self, o):
        return type(o) is type(self) and list(o.coords) == list(self.coords)

    def __hash__(self):
        return hash(tuple(self.coords))


class LinearRing(LineString):
    pass


class Point(BaseGeometry):
    def __init__(self, *xy):
        if len(xy) == 1:
            xy = tuple(np.asarray(xy[0]).ravel())
        self.x, self.y = float(xy[0]), float(xy[1])

    @property
    def xy(self):
        return ([self.x], [self.y])

    @property
    def coords(self):
        return _Coords([(self.x, self.y)])

    def buffer(self, r, quad_segs=16, **k):
        """64-gon inscribed in the circle, vertices at k*pi/32 (GEOS default)."""
        n = 4 * quad_segs
        ang = [2.0 * math.pi * i / n for i in range(n)]
        return Polygon([(self.x + r * math.cos(a), self.y - r * math.sin(a)) for a in ang])

    def __eq__(self, o):
        return type(o) is type(self) and (o.x, o.y) == (self.x, self.y)

    def __hash__(self):
        return hash((self.x, self.y))


class Polygon(BaseGeometry):
    def __init__(self, shell=None, holes=None):
        if isinstance(shell, Polygon):
            holes = shell.interiors if holes is None else holes
            shell = shell.exterior.coords
        if isinstance(shell, LineString):
            shell = shell.coords
        pts = [] if shell is None else [
            tuple(float(v) for v in np.asarray(c).ravel()[:2]) for c in shell
        ]
        if pts and pts[0] != pts[-1]:
            pts.append(pts[0])
        self.exterior = LinearRing(pts)
        self.interiors = [LinearRing(h) for h in (holes or [])]

    # -- helpers -----------------------------------------------------------
    def _ring(self):
        return list(self.exterior.coords[:-1])

    @property
    def area(self):
        c = self._ring()
        if len(c) < 3:
            return 0.0
        s = 0.0
        for (x0, y0), (x1, y1) in zip(c, c[1:] + c[:1]):
            s += x0 * y1 - x1 * y0
        return abs(s) / 2.0

    @property
    def bounds(self):
        xs, ys = self.exterior.coords.xy
        return (min(xs), min(ys), max(xs), max(ys))

    @property
    def centroid(self):
        """Area centroid summed over the triangle fan from the first vertex (the way GEOS accumulates it)."""
        c = self._ring()
        if not c:
            return Point(float("nan"), float("nan"))
        a2 = sx = sy = 0.0
        x0, y0 = c[0]
        for (x1, y1), (x2, y2) in zip(c[1:-1], c[2:]):
            t2 = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0)
            sx += t2 * (x0 + x1 + x2)
            sy += t2 * (y0 + y1 + y2)
            a2 += t2
        if a2 != 0.0:
            return Point(sx / 3 / a2, sy / 3 / a2)
        return Point(sum(p[0] for p in c) / len(c), sum(p[1] for p in c) / len(c))

    def intersection(self, other):
        """Convex rings only (bounding boxes): Sutherland-Hodgman, self clipped by other."""
        A = list(self._ring())
        C = list(other._ring())
        orient = sum(C[k][0] * C[(k + 1) % len(C)][1] - C[(k + 1) % len(C)][0] * C[k][1] for k in range(len(C)))
        sgn = 1.0 if orient >= 0 else -1.0
        for k in range(len(C)):
            if not A:
                break
            (cx, cy), (mx, my) = C[k], C[(k + 1) % len(C)]
            ex, ey = mx - cx, my - cy
            B = []
            for i in range(len(A)):
                (ax, ay), (bx, by) = A[i], A[(i + 1) % len(A)]
                di = sgn * (ex * (ay - cy) - ey * (ax - cx))
                dj = sgn * (ex * (by - cy) - ey * (bx - cx))
                if di >= 0:
                    B.append((ax, ay))
                if (di > 0 and dj < 0) or (di < 0 and dj > 0):
                    u = di / (di - dj)
                    B.append((ax + u * (bx - ax), ay + u * (by - ay)))
            A = B
        return Polygon(A)

    def contains(self, other):
        if isinstance(other, Point):
            return bool(_contains_xy(self, [other.x], [other.y])[0])
        raise NotImplementedError

    def intersects(self, other):
        if isinstance(other, LineString) and not isinstance(other, LinearRing):
            return convex_intersects_segment_exact(self._ring(), list(other.coords))
        return convex_intersects_exact(self._ring(), other._ring())

    def __eq__(self, o):
        return (
            type(o) is type(self)
            and list(o.exterior.coords) == list(self.exterior.coords)
            and len(o.interiors) == len(self.interiors)
        )

    def __hash__(self):
        return hash(tuple(self.exterior.coords))


class MultiPolygon(BaseGeometry):
    def __init__(self, polygons=None):
        self.geoms = list(polygons) if polygons is not None else []

    @property
    def area(self):
        return float(sum(g.area for g in self.geoms))

    def contains(self, other):
        return any(g.contains(other) for g in self.geoms)

    def __eq__(self, o):
        return type(o) is type(self) and o.geoms == self.geoms

    def __hash__(self):
        return hash(tuple(self.geoms))


def _orient(ring):
    s = Fraction(0)
    for (x0, y0), (x1, y1) in zip(ring, ring[1:] + ring[:1]):
        s += _F(x0) * _F(y1) - _F(x1) * _F(y0)
    return 1 if s > 0 else (-1 if s < 0 else 0)


def convex_intersects_exact(A, B):
    """Closed-set intersection of two convex rings, exact rational SAT."""
    # (speed only: rings whose bounding boxes are a clear distance apart cannot touch -- the exact test would say the same)
    ax0, ax1 = min(p[0] for p in A), max(p[0] for p in A)
    ay0, ay1 = min(p[1] for p in A), max(p[1] for p in A)
    bx0, bx1 = min(p[0] for p in B), max(p[0] for p in B)
    by0, by1 = min(p[1] for p in B), max(p[1] for p in B)
    if ax0 - bx1 > 1e-6 or bx0 - ax1 > 1e-6 or ay0 - by1 > 1e-6 or by0 - ay1 > 1e-6:
        return False
    for P, Q in ((A, B), (B, A)):
        o = _orient(P)
        if o == 0:
            continue
        Pf = [(_F(x), _F(y)) for x, y in P]
        Qf = [(_F(x), _F(y)) for x, y in Q]
        for (ax, ay), (bx, by) in zip(Pf, Pf[1:] + Pf[:1]):
            ex, ey = bx - ax, by - ay
            # Q strictly outside this edge's half-plane => separated
            if all(o * (ex * (qy - ay) - ey * (qx - ax)) < 0 for qx, qy in Qf):
                return False
    return True


def _rings(poly):
    out = [list(poly.exterior.coords[:-1])]
    for h in getattr(poly, "interiors", []):
        c = list(h.coords)
        out.append(c[:-1] if len(c) > 1 and c[0] == c[-1] else c)
    return out


def _contains_xy(poly, xs, ys):
    """Strict interior test of points in a polygon with holes: crossing number of the ray towards +x over all rings,
    points on a ring are outside.  numpy fp64 with an error bound on the orientation determinant; whatever falls inside
    the bound is decided in exact rational arithmetic."""
    xs, ys = np.asarray(xs, np.float64).ravel(), np.asarray(ys, np.float64).ravel()
    out = np.zeros(len(xs), dtype=bool)
    if len(xs) == 0 or len(poly.exterior.coords) < 4:
        return out
    rings = _rings(poly)
    E = np.array([(r[i] + r[(i + 1) % len(r)]) for r in rings for i in range(len(r))], np.float64)  # x1 y1 x2 y2
    x0, y0, x1, y1 = poly.bounds
    cand = np.nonzero((xs >= x0) & (xs <= x1) & (ys >= y0) & (ys <= y1))[0]
    for c0 in range(0, len(cand), 4096):
        idx = cand[c0:c0 + 4096]
        px, py = xs[idx, None], ys[idx, None]
        ax, ay, bx, by = E[None, :, 0], E[None, :, 1], E[None, :, 2], E[None, :, 3]
        straddle = ((ay > py) & (by <= py)) | ((by > py) & (ay <= py))
        dl, dr = (ax - px) * (by - py), (ay - py) * (bx - px)
        det = dl - dr
        unsure = straddle & (np.abs(det) <= 1e-14 * (np.abs(dl) + np.abs(dr)))
        on_vertex = ((px == bx) & (py == by)).any(axis=1)
        horiz = ((ay == py) & (by == py) & (px >= np.minimum(ax, bx)) & (px <= np.maximum(ax, bx))).any(axis=1)
        left = np.where(by < ay, -det, det) > 0
        inside = (np.count_nonzero(straddle & left, axis=1) % 2) == 1
        res = inside & ~on_vertex & ~horiz
        for k in np.nonzero(unsure.any(axis=1) | on_vertex | horiz)[0]:
            res[k] = _contains_exact(rings, float(px[k, 0]), float(py[k, 0]))
        out[idx] = res
    return out


def _contains_exact(rings, x, y):
    px, py = _F(x), _F(y)
    cross = 0
    for r in rings:
        R = [(_F(a), _F(b)) for a, b in r]
        for (ax, ay), (bx, by) in zip(R, R[1:] + R[:1]):
            if (px, py) == (bx, by):
                return False
            if ay == py and by == py:
                if min(ax, bx) <= px <= max(ax, bx):
                    return False
                continue
            if (ay > py and by <= py) or (by > py and ay <= py):
                det = (ax - px) * (by - py) - (ay - py) * (bx - px)
                if det == 0:
                    return False
                if by < ay:
                    det = -det
                if det > 0:
                    cross += 1
    return cross % 2 == 1


def convex_intersects_segment_exact(ring, seg):
    """Closed convex ring against a closed 2-point segment, exact rational: an endpoint inside or on the ring, or the
    segment meeting one of its edges."""
    (ax, ay), (bx, by) = [(_F(x), _F(y)) for x, y in seg[:2]]
    R = [(_F(x), _F(y)) for x, y in ring]

    def orient(p, q, r):
        v = (q[0] - p[0]) * (r[1] - p[1]) - (q[1] - p[1]) * (r[0] - p[0])
        return (v > 0) - (v < 0)

    def inside_closed(p):
        signs = [orient(R[k], R[(k + 1) % len(R)], p) for k in range(len(R))]
        return not (any(s > 0 for s in signs) and any(s < 0 for s in signs))

    def on_seg(p, q, r):
        return min(p[0], q[0]) <= r[0] <= max(p[0], q[0]) and min(p[1], q[1]) <= r[1] <= max(p[1], q[1])

    A, B = (ax, ay), (bx, by)
    if inside_closed(A) or inside_closed(B):
        return True
    for k in range(len(R)):
        Cc, D = R[k], R[(k + 1) % len(R)]
        o1, o2, o3, o4 = orient(A, B, Cc), orient(A, B, D), orient(Cc, D, A), orient(Cc, D, B)
        if o1 * o2 < 0 and o3 * o4 < 0:
            return True
        if (o1 == 0 and on_seg(A, B, Cc)) or (o2 == 0 and on_seg(A, B, D)) or (o3 == 0 and on_seg(Cc, D, A)) or \
                (o4 == 0 and on_seg(Cc, D, B)):
            return True
    return False
