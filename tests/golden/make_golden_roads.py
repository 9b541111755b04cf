#!/usr/bin/env python3
"""Golden vectors for road surfaces (N2 map layers, N4 `ego_off_road`) from the REAL reference:
tests/golden/roads.npz.

Build container only (needs /root/reference and the import stand-ins of tests/golden/_refstubs, see its README):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden_roads.py

What comes from the reference: which geometries a road network holds after `RoadNetwork.create_from_json`, their
boundary rings, which of them each union takes (`driveable_surface`, `walkable_surface`, the per-layer unions of
RasterizedMapSensor), the raster coordinates, the gym loop and its terminal condition.  What does not: shapely's
`contains`.  The stand-in answers it with the crossing number of the rings, decided in exact rational arithmetic
wherever fp64 could be in doubt -- the mathematical answer for the given coordinates (GEOS computes the same predicate
with robust orientation tests), except on edges shared by two polygons of one union, where GEOS (which dissolves them)
says inside and the stand-in outside.  No sampled point lies on such an edge unless noted by `n_on_edge` below.
Only data is stored.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path[:0] = [os.path.join(HERE, "_refstubs"), "/root/reference"]

import numpy as np  # noqa: E402

import scenario_gym  # noqa: E402
from scenario_gym import ScenarioGym  # noqa: E402
from scenario_gym.road_network import RoadNetwork  # noqa: E402
from scenario_gym.sensor.map import RasterizedMapSensor  # noqa: E402
from scenario_gym.trajectory import Trajectory  # noqa: E402
from scenario_gym.xosc_interface import import_scenario  # noqa: E402
from shapely.geometry import Point  # noqa: E402  (stand-in)
from shapely.ops import unary_union  # noqa: E402  (stand-in)

assert scenario_gym.__version__ == "0.3.1"
SCEN_DIR = "/root/reference/tests/input_files/Scenarios"
NET_DIR = "/root/reference/tests/input_files/Road_Networks"
ETYPE = {"Vehicle": 0, "Pedestrian": 1}
LAYERS = RasterizedMapSensor._all_layers  # entity, driveable_surface, road, intersection, lane, walkable_surface, ...
BITS = dict(driveable_surface=1, road=2, intersection=4, lane=8, walkable_surface=16, pavement=32, crossing=64)


def export_network(rn):
    """Rings + layer bits, taken from the reference's objects (the unions exactly as the reference composes them)."""
    member = {
        1: [g for g in rn.road_network_geometries if g.driveable],
        2: list(rn.roads), 4: list(rn.intersections), 8: [l for r in rn.roads for l in r.lanes],
        16: [g for g in rn.road_network_geometries if g.walkable],
        32: list(rn.pavements), 64: list(rn.crossings),
        128: [g for g in rn.road_network_geometries if g.impenetrable],
    }
    geoms = {}
    for bit, gs in member.items():
        for g in gs:
            geoms.setdefault(g.id, [g, 0])[1] |= bit
    rings, ring_off, layers = [], [0], []
    # sorted by geometry id: the reference walks some of its collections in an order that changes from one interpreter run
    # to the next, and a fixture must come out the same every time it is regenerated
    ids = sorted(geoms)
    for g, bits in (geoms[k] for k in ids):
        rs = [np.array(g.boundary.exterior.coords)[:-1]] + [np.array(i.coords) for i in g.boundary.interiors]
        rings += rs
        ring_off.append(len(rings))
        layers.append(bits)
    vert_off = np.concatenate([[0], np.cumsum([len(r) for r in rings])]).astype(np.int64)
    return dict(ring_off=np.array(ring_off, np.int64), vert_off=vert_off, verts=np.concatenate(rings, axis=0),
                layers=np.array(layers, np.uint32), ids=np.array(ids))


def export_scenario(out, key, s):
    ents = s.entities
    off = np.concatenate([[0], np.cumsum([e.trajectory.data.shape[0] for e in ents])]).astype(np.int64)
    out[f"{key}/scenario/knot_off"] = off
    out[f"{key}/scenario/knots"] = np.concatenate([e.trajectory.data for e in ents], axis=0)
    out[f"{key}/scenario/bbox"] = np.array([[e.bounding_box.width, e.bounding_box.length, e.bounding_box.center_x,
                                             e.bounding_box.center_y] for e in ents], np.float64)
    out[f"{key}/scenario/etype"] = np.array([ETYPE.get(e.catalog_entry.catalog_type, 2) for e in ents], np.int32)
    out[f"{key}/scenario/refs"] = np.array([e.ref for e in ents])
    out[f"{key}/scenario/ego"] = np.int64(ents.index(s.ego))
    out[f"{key}/scenario/length"] = np.float64(s.length)


def surfaces(rn):
    return dict(driveable_surface=rn.driveable_surface, walkable_surface=rn.walkable_surface,
                road=unary_union([r.boundary for r in rn.roads]),
                intersection=unary_union([i.boundary for i in rn.intersections]),
                lane=unary_union([l.boundary for r in rn.roads for l in r.lanes]),
                pavement=unary_union([p.boundary for p in rn.pavements]),
                crossing=unary_union([c.boundary for c in rn.crossings]))


def main():
    out = {}
    rng = np.random.default_rng(20240807)
    nets = sorted(f[:-5] for f in os.listdir(NET_DIR) if f.endswith(".json"))
    out["networks"] = np.array(nets)
    for n in nets:
        rn = RoadNetwork.create_from_json(os.path.join(NET_DIR, n + ".json"))
        for k, v in export_network(rn).items():
            out[f"net/{n}/{k}"] = v
        # known answers: random points over the network, points on vertices / edge midpoints / just beside them
        V = out[f"net/{n}/verts"]
        lo, hi = V.min(0) - 5, V.max(0) + 5
        pts = [rng.uniform(lo, hi, (1500, 2))]
        pick = V[rng.integers(0, len(V), 200)]
        pts += [pick, pick + rng.normal(0, 1e-9, pick.shape), pick + rng.normal(0, 0.5, pick.shape)]
        i = rng.integers(0, len(V) - 1, 200)
        pts.append((V[i] + V[i + 1]) / 2)
        pts = np.concatenate(pts)
        out[f"net/{n}/points"] = pts
        for name, surf in surfaces(rn).items():
            out[f"net/{n}/contains_{name}"] = np.array([surf.contains(Point(x, y)) for x, y in pts], np.uint8)
        print(n, len(out[f"net/{n}/layers"]), {k: int(out[f"net/{n}/contains_{k}"].sum()) for k in BITS})

    # ---- RasterizedMapSensor, all layers, along the reference's own rollouts (tests/test_sensor.py:38-77) ----
    # one scenario per road network of the shipped scenarios (6-lane intersection, roundabout, rural, Greenwich with
    # pavements and crossings)
    names = ["a5e43fe4-646a-49ba-82ce-5f0063776566", "41dac6fa-6f83-461e-a145-08692da5f3c7",
             "a98d5c7d-76aa-49bf-b88c-97db5d5c7433", "3fee6507-fd24-432f-b781-ca5676c834ef"]
    out["scenarios"] = np.array(names)
    out["layers"] = np.array(LAYERS)
    out["raster_cfg"] = np.array([[30.0, 30.0, 61.0], [20.0, 20.0, 20.0]])  # width, height, n (the test's and the default)
    for n in names:
        s = import_scenario(os.path.join(SCEN_DIR, n + ".xosc"))
        export_scenario(out, n, s)
        out[f"{n}/network"] = np.array(s.road_network.name)
        gym = ScenarioGym(timestep=0.1)
        gym.set_scenario(s)
        ego = gym.state.scenario.entities[0]
        sensors = [RasterizedMapSensor(ego, layers=LAYERS, width=w, height=h, freq=None, n=int(k), channels_first=True)
                   for w, h, k in out["raster_cfg"]]
        maps = [[np.asarray(r.reset(gym.state).map)] for r in sensors]
        steps, k = [0], 0
        while not gym.state.is_done:
            gym.step()
            k += 1
            if k % 30 == 0:
                steps.append(k)
                for m, r in zip(maps, sensors):
                    m.append(np.asarray(r.step(gym.state).map))
        out[f"{n}/map_steps"] = np.array(steps)
        for c, m in enumerate(maps):
            out[f"{n}/map{c}"] = np.array(m, np.uint8)  # [frames][layer][n][n] as the sensor lays it out (channels_first)
        print(n, s.road_network.name, len(steps), [np.array(m).sum(axis=(0, 2, 3)).tolist() for m in maps])

    # ---- ego_off_road terminal condition (state/state.py:401-407) ----
    # the recorded egos stay on the road; a copy whose ego drifts sideways leaves it mid-way
    for n in names:
        for tag, drift in (("onroad", 0.0), ("drift", 0.6)):
            s = import_scenario(os.path.join(SCEN_DIR, n + ".xosc"))
            if drift:
                d = s.ego.trajectory.data.copy()
                h0 = d[0, 4]
                d[:, 1] += -np.sin(h0) * drift * (d[:, 0] - d[0, 0])
                d[:, 2] += np.cos(h0) * drift * (d[:, 0] - d[0, 0])
                s.ego.trajectory = Trajectory(d)
            gym = ScenarioGym(timestep=0.1, terminal_conditions=["max_length", "ego_off_road"])
            gym.set_scenario(s)
            key = f"{n}/{tag}"
            export_scenario(out, key, s)
            ts = [gym.state.t]
            while not gym.state.is_done:
                gym.step()
                ts.append(gym.state.t)
            out[f"{key}/t"] = np.array(ts)
            out[f"{key}/final_ego"] = np.array(gym.state.poses[s.ego])
            print(key, len(ts) - 1, ts[-1], "length", s.length)
    np.savez_compressed(os.path.join(os.environ.get("SG_GOLDEN_OUT", HERE), "roads.npz"), **out)


if __name__ == "__main__":
    main()
