import numpy as np

from .geometry.base import MultiPolygon, _contains_xy


def contains(area, x, y):
    x, y = np.atleast_1d(x), np.atleast_1d(y)
    if isinstance(area, MultiPolygon):
        out = np.zeros(len(x), dtype=bool)
        for g in area.geoms:
            out |= _contains_xy(g, x, y)
        return out
    return _contains_xy(area, x, y)
