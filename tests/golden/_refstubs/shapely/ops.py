from .geometry.base import MultiPolygon, Polygon


def unary_union(geoms):
    out = []
    for g in geoms:
        if isinstance(g, MultiPolygon):
            out.extend(g.geoms)
        elif isinstance(g, Polygon):
            out.append(g)
    return MultiPolygon(out)


def nearest_points(a, b):
    """GEOS DistanceOp for (areal geometry, Point), restated: a point inside or on the geometry is its own nearest point;
    otherwise every ring segment in order, Distance::pointToSegment picks the nearest (first on ties),
    LineSegment::closestPoint gives the point.  Plain fp64, the formulas of the published JTS / GEOS sources."""
    import math

    from .geometry.base import Point, _rings

    geoms = a.geoms if isinstance(a, MultiPolygon) else [a]
    px, py = b.x, b.y
    for g in geoms:
        if g.contains(b) or _on_boundary(g, px, py):
            return Point(px, py), b

    def dist(ax, ay, bx, by):
        return math.sqrt((ax - bx) * (ax - bx) + (ay - by) * (ay - by))

    best, close = math.inf, (px, py)
    for g in geoms:
        for ring in _rings(g):
            for (ax, ay), (bx, by) in zip(ring, ring[1:] + ring[:1]):
                if (ax, ay) == (bx, by):
                    d = dist(px, py, ax, ay)
                else:
                    len2 = (bx - ax) * (bx - ax) + (by - ay) * (by - ay)
                    r = ((px - ax) * (bx - ax) + (py - ay) * (by - ay)) / len2
                    if r <= 0.0:
                        d = dist(px, py, ax, ay)
                    elif r >= 1.0:
                        d = dist(px, py, bx, by)
                    else:
                        d = abs(((ay - py) * (bx - ax) - (ax - px) * (by - ay)) / len2) * math.sqrt(len2)
                if d < best:
                    best = d
                    if (px, py) == (ax, ay):
                        f = 0.0
                    elif (px, py) == (bx, by):
                        f = 1.0
                    else:
                        dx, dy = bx - ax, by - ay
                        ln = dx * dx + dy * dy
                        f = math.nan if ln <= 0.0 else ((px - ax) * dx + (py - ay) * dy) / ln
                    if 0.0 < f < 1.0:
                        close = (ax + f * (bx - ax), ay + f * (by - ay))
                    else:
                        close = (ax, ay) if dist(ax, ay, px, py) < dist(bx, by, px, py) else (bx, by)
    return Point(*close), b


def _on_boundary(g, px, py):
    from .geometry.base import _contains_exact, _rings

    # inside-or-on = not strictly outside: strictly inside is handled by contains(); on the boundary <=> neither the
    # point nor ... decided exactly: the crossing test reports boundary points as "not contained", so test the edges
    from fractions import Fraction as F

    P = (F(px), F(py))
    for ring in _rings(g):
        R = [(F(x), F(y)) for x, y in ring]
        for (ax, ay), (bx, by) in zip(R, R[1:] + R[:1]):
            cr = (bx - ax) * (P[1] - ay) - (by - ay) * (P[0] - ax)
            if cr == 0 and min(ax, bx) <= P[0] <= max(ax, bx) and min(ay, by) <= P[1] <= max(ay, by):
                return True
    return False
