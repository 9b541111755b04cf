"""Routes for pedestrian agents along the walkable part of a road network: pedestrian/route.py.

Host-side scenario setup (the device consumes the finished routes as PedestrianAgent.route): nodes every ~1 m along the
centre lines of pavements and crossings, edges between consecutive nodes and between a crossing and the pavements it
connects (at their closest nodes), breadth-first shortest paths.
"""
import random
from itertools import chain
from typing import Dict, List, Optional, Tuple

import numpy as np

from .road_network import RoadNetwork


def _interpolate(line: np.ndarray, s: float) -> np.ndarray:
    """LineString.interpolate(s, normalized=False): the point at arc length s (clamped to the ends)."""
    seg = np.diff(line, axis=0)
    length = np.sqrt((seg ** 2).sum(axis=1))
    if s <= 0.0 or len(line) == 1:
        return line[0].copy()
    acc = 0.0
    for k, L in enumerate(length):
        if s < acc + L or (k == len(length) - 1 and s <= acc + L):
            u = 0.0 if L == 0.0 else (s - acc) / L
            return line[k] + u * seg[k]
        acc += L
    return line[-1].copy()


def _center_nodes(center: np.ndarray) -> np.ndarray:
    total = float(np.sqrt((np.diff(center, axis=0) ** 2).sum(axis=1)).sum())
    return np.array([_interpolate(center, x) for x in np.linspace(0.0, total, int(total))]).reshape(-1, 2)


def make_pedestrian_connection_graph(rn: RoadNetwork):
    """route.py:53-128: (graph {node: [neighbours]}, node_to_idx {"<object id>_<i>": node}, node_data {node: (x, y)})."""
    graph: Dict[int, List[int]] = {}
    node_to_idx: Dict[str, int] = {}
    node_data: Dict[int, Tuple[float, float]] = {}
    pavement_coords = {p.id: _center_nodes(p.center) for p in rn.pavements}
    crossing_coords = {c.id: _center_nodes(c.center) for c in rn.crossings}
    for obj, coords in chain(pavement_coords.items(), crossing_coords.items()):
        for i, (x, y) in enumerate(coords):
            node_to_idx[f"{obj}_{i}"] = len(node_to_idx)
            graph[node_to_idx[f"{obj}_{i}"]] = []
            node_data[node_to_idx[f"{obj}_{i}"]] = (x, y)
    for obj, coords in chain(pavement_coords.items(), crossing_coords.items()):
        for i in range(len(coords) - 1):
            graph[node_to_idx[f"{obj}_{i}"]].append(node_to_idx[f"{obj}_{i + 1}"])
            graph[node_to_idx[f"{obj}_{i + 1}"]].append(node_to_idx[f"{obj}_{i}"])
    for c in rn.crossings:
        for p in c.pavements:
            c_coords, p_coords = crossing_coords[c.id], pavement_coords[p]
            c_idx, p_idx = np.unravel_index(
                np.linalg.norm(c_coords[:, None, :] - p_coords[None, :, :], axis=-1).argmin(),
                (c_coords.shape[0], p_coords.shape[0]))
            graph[node_to_idx[f"{c.id}_{c_idx}"]].append(node_to_idx[f"{p}_{p_idx}"])
            graph[node_to_idx[f"{p}_{p_idx}"]].append(node_to_idx[f"{c.id}_{c_idx}"])
    return graph, node_to_idx, node_data


def shortest_path(graph: Dict[int, List[int]], start: int, goal: int) -> Optional[List[int]]:
    """route.py:131-158: breadth-first search; None when start and goal are not connected."""
    if start == goal:
        return [start]
    explored, queue = set(), [[start]]
    while queue:
        path = queue.pop(0)
        node = path[-1]
        if node not in explored:
            for neighbour in graph[node]:
                new_path = path + [neighbour]
                queue.append(new_path)
                if neighbour == goal:
                    return new_path
            explored.add(node)
    return None


def find_route(graph, node_data, start: np.ndarray, finish: np.ndarray) -> Optional[np.ndarray]:
    """route.py:161-190: start, the shortest node path between the nodes closest to start and finish, finish."""
    if not node_data:
        return np.array([start] + [finish])
    start_node = min(node_data, key=lambda n: np.linalg.norm(np.array(node_data[n]) - start))
    end_node = min(node_data, key=lambda n: np.linalg.norm(np.array(node_data[n]) - finish))
    route = shortest_path(graph, start_node, end_node)
    if route is None:
        return None
    return np.array([start] + [list(node_data[n]) for n in route] + [finish])


class RouteFinder:
    """route.py:10-50."""

    def __init__(self, rn: RoadNetwork):
        self.rn = rn
        self.graph, self.node_to_idx, self.node_data = make_pedestrian_connection_graph(rn)

    def find_route(self, start: np.ndarray, finish: np.ndarray) -> Optional[np.ndarray]:
        return find_route(self.graph, self.node_data, np.asarray(start, float), np.asarray(finish, float))

    def generate_route(self, n: int, start: Optional[np.ndarray] = None, no_repeat: bool = False):
        if start is not None:
            route = [min(self.node_data, key=lambda x: np.linalg.norm(self.node_data[x] - start))]
        else:
            route = [random.choice(list(self.graph.keys()))]
        while len(route) < n:
            suc = self.graph[route[-1]]
            if no_repeat:
                suc = list(set(suc).difference(route))
            if not suc:
                break
            route.append(random.choice(suc))
        return [self.node_data[i] for i in route]
