import numpy as np


class STRtree:
    def __init__(self, geoms):
        self.geometries = np.empty(len(geoms), dtype=object)
        for i, g in enumerate(geoms):
            self.geometries[i] = g

    def query(self, g, predicate=None):
        assert predicate == "intersects"
        return np.array(
            [i for i, o in enumerate(self.geometries) if g.intersects(o)], dtype=np.intp
        )
