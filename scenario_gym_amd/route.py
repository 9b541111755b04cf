"""Pedestrian routing over the walkable part of a road network (host-side scenario setup).

Mirrors the behaviour of the reference's `scenario_gym/pedestrian/route.py:10-190`: sample points roughly every metre
along the centre line of every pavement and crossing, connect consecutive samples, connect each crossing to the
pavements it lists at their mutually closest samples, and answer "route from a to b" with a breadth-first search between
the samples closest to a and b.  The finished routes are what the device consumes (`PedestrianAgent.route`).

The walk graph is kept as arrays: sample coordinates `xy[n, 2]`, string labels `"<object id>_<k>"`, and a CSR
adjacency (`adj_off`, `adj`) whose per-node neighbour order is the insertion order the reference's dict-of-lists ends up
with -- breadth-first tie-breaking depends on it.  The dict views the reference exposes (`graph`, `node_to_idx`,
`node_data`) are derived from the arrays on demand.
"""
import random
from typing import Dict, List, Optional, Tuple

import numpy as np

from .road_network import RoadNetwork


def sample_centre_line(line: np.ndarray) -> np.ndarray:
    """Points at arc lengths np.linspace(0, L, int(L)) of a polyline (shapely `interpolate(s, normalized=False)`)."""
    line = np.asarray(line, dtype=float).reshape(-1, 2)
    step = np.diff(line, axis=0)
    seg_len = np.hypot(step[:, 0], step[:, 1]) if len(step) else np.zeros(0)
    cum = np.concatenate([[0.0], np.cumsum(seg_len)])
    total = float(seg_len.sum())
    s = np.linspace(0.0, total, int(total))
    if s.size == 0:
        return np.zeros((0, 2))
    if len(line) == 1:
        return np.repeat(line[:1], s.size, axis=0)
    # segment holding each arc length: the first one whose end lies beyond s (the last segment also takes s == total)
    seg = np.clip(np.searchsorted(cum, s, side="right") - 1, 0, len(seg_len) - 1)
    with np.errstate(invalid="ignore", divide="ignore"):
        frac = np.where(seg_len[seg] > 0.0, (s - cum[seg]) / seg_len[seg], 0.0)
    pts = line[seg] + frac[:, None] * step[seg]
    pts[s <= 0.0] = line[0]
    return pts


class WalkGraph:
    """Array form of the pedestrian connection graph (reference `make_pedestrian_connection_graph`, route.py:53-128)."""

    def __init__(self, rn: RoadNetwork):
        strips = [(p.id, sample_centre_line(p.center)) for p in rn.pavements]
        n_pavements = len(strips)
        strips += [(c.id, sample_centre_line(c.center)) for c in rn.crossings]
        first: Dict[str, int] = {}  # object id -> index of its first sample
        labels: List[str] = []
        coords = []
        for oid, pts in strips:
            first[oid] = len(labels)
            labels.extend(f"{oid}_{k}" for k in range(len(pts)))
            coords.append(pts)
        self.labels = labels
        self.xy = np.concatenate(coords, axis=0) if coords else np.zeros((0, 2))
        count = {oid: len(pts) for oid, pts in strips}
        # directed edges in the order the reference appends them
        src: List[np.ndarray] = []
        dst: List[np.ndarray] = []
        for oid, pts in strips:
            if len(pts) > 1:
                a = first[oid] + np.arange(len(pts) - 1)
                src.append(np.stack([a, a + 1], axis=1).ravel())
                dst.append(np.stack([a + 1, a], axis=1).ravel())
        by_id = dict(strips)
        for crossing in rn.crossings:
            cpts = by_id[crossing.id]
            for pid in crossing.pavements:
                ppts = by_id[pid]
                gap = np.linalg.norm(cpts[:, None, :] - ppts[None, :, :], axis=-1)
                ci, pi = divmod(int(gap.argmin()), count[pid])  # row-major argmin == np.unravel_index
                u, v = first[crossing.id] + ci, first[pid] + pi
                src.append(np.array([u, v]))
                dst.append(np.array([v, u]))
        n = len(labels)
        s = np.concatenate(src) if src else np.zeros(0, np.int64)
        d = np.concatenate(dst) if dst else np.zeros(0, np.int64)
        order = np.argsort(s, kind="stable")  # per-node neighbour order = insertion order
        self.adj = d[order].astype(np.int64)
        self.adj_off = np.concatenate([[0], np.cumsum(np.bincount(s.astype(np.int64), minlength=n))]).astype(np.int64)
        self.n_pavement_strips = n_pavements

    def __len__(self) -> int:
        return len(self.labels)

    def neighbours(self, node: int) -> np.ndarray:
        return self.adj[self.adj_off[node]:self.adj_off[node + 1]]

    def nearest(self, point) -> int:
        """Index of the sample closest to `point` (first one on ties, like min() over the reference's dict)."""
        d = self.xy - np.asarray(point, dtype=float)[None, :]
        return int(np.argmin(np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1])))

    def bfs_path(self, start: int, goal: int) -> Optional[List[int]]:
        """Breadth-first path with parent pointers, O(V + E).  Equal to the reference's list-of-paths search
        (route.py:131-158): nodes are expanded in the order of their first discovery, neighbours in adjacency order, and
        the search ends the moment the goal is discovered."""
        if start == goal:
            return [start]
        parent = np.full(len(self), -1, np.int64)
        parent[start] = start
        frontier = [start]
        adj, off = self.adj, self.adj_off
        while frontier:
            nxt: List[int] = []
            for u in frontier:
                for v in adj[off[u]:off[u + 1]]:
                    v = int(v)
                    if parent[v] >= 0:
                        continue
                    parent[v] = u
                    if v == goal:
                        path = [v]
                        while path[-1] != start:
                            path.append(int(parent[path[-1]]))
                        return path[::-1]
                    nxt.append(v)
            frontier = nxt
        return None

    # ---- the reference's dict views ------------------------------------------------------------------
    def as_dicts(self) -> Tuple[Dict[int, List[int]], Dict[str, int], Dict[int, Tuple[float, float]]]:
        graph = {i: [int(v) for v in self.neighbours(i)] for i in range(len(self))}
        node_to_idx = {lab: i for i, lab in enumerate(self.labels)}
        node_data = {i: (float(self.xy[i, 0]), float(self.xy[i, 1])) for i in range(len(self))}
        return graph, node_to_idx, node_data


def make_pedestrian_connection_graph(rn: RoadNetwork):
    """(graph, node_to_idx, node_data) in the reference's dict form."""
    return WalkGraph(rn).as_dicts()


def find_route(walk: WalkGraph, start: np.ndarray, finish: np.ndarray) -> Optional[np.ndarray]:
    """start, the samples of the breadth-first path between the samples nearest to start and finish, finish; None when
    they are not connected; the straight pair when the network has no walkable strips (reference route.py:161-190)."""
    start, finish = np.asarray(start, dtype=float), np.asarray(finish, dtype=float)
    if len(walk) == 0:
        return np.stack([start, finish])
    nodes = walk.bfs_path(walk.nearest(start), walk.nearest(finish))
    if nodes is None:
        return None
    return np.concatenate([start[None, :], walk.xy[nodes], finish[None, :]], axis=0)


class RouteFinder:
    """Routes along the walkable areas of a road network (reference `RouteFinder`, route.py:10-50)."""

    def __init__(self, rn: RoadNetwork):
        self.rn = rn
        self.walk = WalkGraph(rn)
        self.graph, self.node_to_idx, self.node_data = self.walk.as_dicts()

    def find_route(self, start: np.ndarray, finish: np.ndarray) -> Optional[np.ndarray]:
        return find_route(self.walk, start, finish)

    def generate_route(self, n: int, start: Optional[np.ndarray] = None, no_repeat: bool = False):
        """Random walk of at most n samples from the sample nearest to `start` (or a random one)."""
        walk = self.walk
        here = walk.nearest(start) if start is not None else random.randrange(len(walk))
        visited = [here]
        while len(visited) < n:
            options = [int(v) for v in walk.neighbours(here)]
            if no_repeat:
                options = list(set(options).difference(visited))
            if not options:
                break
            here = random.choice(options)
            visited.append(here)
        return [self.node_data[i] for i in visited]
