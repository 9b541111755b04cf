#!/usr/bin/env python3
"""Golden vectors for SURVEY 8(f) N1's remaining pieces, from the REAL reference (build container only):

  json.npz       Scenario.to_dict / to_json / from_json (scenario/scenario.py:186-319) of shipped scenarios: everything but the
                 trajectories as one small JSON string per scenario, the trajectories as arrays; the embedded road network
                 (road_network_path=None) as per-object-list digests
  elevation.npz  RoadNetwork.elevation_at_point (road_network/road_network.py:446-520) on a synthetic network that HAS
                 elevation samples, and the reference's import (xosc_interface/read.py:205-217) of a scenario on that network
                 whose vertices lack z: the knots it ends up with and the replayed poses

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden_json.py

Only DATA is written (numbers, names, the synthetic inputs this script generates itself)."""
import hashlib
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path[:0] = [os.path.join(HERE, "_refstubs"), "/root/reference"]

import numpy as np  # noqa: E402

import scenario_gym  # noqa: E402
from scenario_gym import ScenarioGym  # noqa: E402
from scenario_gym.road_network import RoadNetwork  # noqa: E402
from scenario_gym.scenario import Scenario  # noqa: E402
from scenario_gym.xosc_interface import import_scenario  # noqa: E402

assert scenario_gym.__version__ == "0.3.1"
SCEN_DIR = "/root/reference/tests/input_files/Scenarios"
JSON_NAMES = ["a5e43fe4-646a-49ba-82ce-5f0063776566", "3fee6507-fd24-432f-b781-ca5676c834ef", "41dac6fa-6f83-461e-a145-08692da5f3c7",
              "5c5188e0-715a-4dd2-a6b2-b3c96b52d608", "a98d5c7d-76aa-49bf-b88c-97db5d5c7433", "mixed_catalogs", "no_references",
              "3e39a079-5653-440c-bcbe-24dc9f6bf0e6", "1518e754-318f-4847-8a30-2dce552b4504"]


def split(d):
    """to_dict() -> (JSON text without the trajectories, list of trajectory arrays)."""
    d = json.loads(json.dumps(d))
    trajs = []
    for e in d["entities"]:
        trajs.append(np.array(e["trajectory"], np.float64))
        e["trajectory"] = len(trajs) - 1
    return json.dumps(d, sort_keys=True), trajs


def canonical(o):
    """Lists the reference builds through a Python set (a lane's successors / predecessors, the network's lanes) come out in
    a per-process order: sorted here."""
    if isinstance(o, dict):
        return {k: (sorted(v) if k in ("successors", "predecessors") else canonical(v)) for k, v in o.items()}
    if isinstance(o, list):
        return [canonical(v) for v in o]
    return o


def digest_list(objs):
    """Order-independent digest of a list of road-object dicts."""
    return hashlib.sha256(json.dumps(sorted(canonical(objs), key=lambda o: o["id"]), sort_keys=True).encode()).hexdigest()


def make_json():
    out = {}
    names = [n for n in JSON_NAMES if os.path.exists(os.path.join(SCEN_DIR, n + ".xosc"))]
    assert len(names) == len(JSON_NAMES), names
    for n in names:
        s = import_scenario(os.path.join(SCEN_DIR, n + ".xosc"))
        meta, trajs = split(s.to_dict())
        out[f"{n}/to_dict"] = np.array(meta)
        for i, t in enumerate(trajs):
            out[f"{n}/traj_{i}"] = t
        with tempfile.TemporaryDirectory() as tmp:
            os.makedirs(os.path.join(tmp, "Scenarios"))
            p = os.path.join(tmp, "Scenarios", "s.json")
            # the road network by path: the shipped directory, so that from_json finds the file again
            rn_dir = "/root/reference/tests/input_files/Road_Networks"
            s.to_json(p, road_network_path=rn_dir)
            back = Scenario.from_json(p)
            meta2, trajs2 = split(back.to_dict(road_network_path=rn_dir))
            out[f"{n}/roundtrip"] = np.array(meta2)
            assert all(np.array_equal(a, b) for a, b in zip(trajs, trajs2))
            out[f"{n}/roundtrip_classes"] = np.array([type(e).__name__ for e in back.entities])
            out[f"{n}/roundtrip_has_network"] = np.array(back.road_network is not None)
        if s.road_network is not None and n == names[0]:
            emb = s.to_dict(road_network_path=None)["road_network"]
            out[f"{n}/embedded_keys"] = np.array(sorted(emb.keys()))
            for k, v in emb.items():
                if isinstance(v, list):
                    out[f"{n}/embedded/{k}"] = np.array(digest_list(v))
                    out[f"{n}/embedded_n/{k}"] = np.array(len(v))
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(os.environ.get("SG_GOLDEN_OUT", HERE), "json.npz"), **out)
    print("json.npz:", len(out), "arrays,", names)


def synthetic_network(rng):
    """A 3 x 2 grid of square road tiles over [0, 60] x [0, 40], every road and the intersection with (x, y, z) samples of a
    tilted, gently curved surface; no lanes (their order in the reference is per-process)."""
    def surf(x, y):
        return 0.05 * x - 0.02 * y + 0.001 * x * y + 2.0

    def tile(i, x0, y0, w=20.0, h=20.0, n=18):
        xs, ys = rng.uniform(x0, x0 + w, n), rng.uniform(y0, y0 + h, n)
        ring = [(x0, y0), (x0 + w, y0), (x0 + w, y0 + h), (x0, y0 + h)]
        return {"id": f"r{i}", "Boundary": [{"x": x, "y": y} for x, y in ring],
                "Center": [{"x": x0, "y": y0 + h / 2}, {"x": x0 + w, "y": y0 + h / 2}], "Lanes": [],
                "Elevation": [[float(x), float(y), float(surf(x, y))] for x, y in zip(xs, ys)]}

    roads = [tile(i, 20.0 * (i % 3), 20.0 * (i // 3)) for i in range(5)]
    inter = tile(5, 40.0, 20.0)
    inter["id"] = "i0"
    del inter["Center"]
    inter["connecting_roads"] = ["r2", "r4"]
    return {"name": "synthetic_elevation", "Roads": roads, "Intersections": [inter]}


def make_elevation():
    sys.path.insert(0, ROOT)
    from scenario_gym_amd import xosc_write as W  # (our own writer: the input file is this script's to generate)

    rng = np.random.default_rng(20240807)
    net = synthetic_network(rng)
    out = {"network_json": np.array(json.dumps(net))}
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "Road_Networks"))
        os.makedirs(os.path.join(tmp, "Scenarios"))
        W.write_catalog(os.path.join(tmp, "Catalogs"))
        npath = os.path.join(tmp, "Road_Networks", "synthetic_elevation.json")
        with open(npath, "w") as f:
            json.dump(net, f)
        rn = RoadNetwork.create_from_json(npath)
        q = np.column_stack([rng.uniform(-15, 75, 400), rng.uniform(-15, 55, 400)])  # inside and outside the samples' hull
        out["query_xy"] = q
        out["query_z"] = np.asarray(rn.elevation_at_point(q[:, 0], q[:, 1]), np.float64)
        out["scalar_z"] = np.asarray(rn.elevation_at_point(31.5, 12.25), np.float64)
        out["broadcast_z"] = np.asarray(rn.elevation_at_point(31.5, q[:5, 1]), np.float64)
        # a scenario on that network: ego without any z, one entity with z at every vertex (kept), one with z at some (all
        # of its z replaced), one static entity (Init teleport only: never filled)
        ents = W.synthetic_entities(rng, 4, 12, duration=6.0, extent=1.0)
        for k, (name, entry, kn) in enumerate(ents):
            kn[:, 1] = 8.0 + 44.0 * (kn[:, 0] - kn[0, 0]) / 6.0 + 3.0 * k      # through the tiles, partly outside the hull
            kn[:, 2] = 5.0 + 6.0 * k + 2.0 * np.sin(kn[:, 0])
        ents[1][2][:, 3] = 1.25
        ents[2][2][::2, 3] = -0.5
        ents[3] = (ents[3][0], ents[3][1], ents[3][2][:1])
        spath = os.path.join(tmp, "Scenarios", "elev.xosc")
        W.write_scenario(spath, ents, road_network_file="../Road_Networks/synthetic_elevation.json")
        with open(spath) as f:
            out["scenario_xosc"] = np.array(f.read())   # (generated by this script with our writer: an input, not reference text)
        s = import_scenario(spath)
        assert s.road_network is not None
        for i, e in enumerate(s.entities):
            out[f"knots_{i}"] = e.trajectory.data
        gym = ScenarioGym(timestep=0.1)
        gym.load_scenario(spath)
        gym.rollout()
        rec = gym.state.recorded_poses()
        out["n_entities"] = np.array(len(s.entities))
        for i, e in enumerate(gym.state.scenario.entities):
            out[f"recorded_{i}"] = np.asarray(rec[e])
    np.savez_compressed(os.path.join(os.environ.get("SG_GOLDEN_OUT", HERE), "elevation.npz"), **out)
    print("elevation.npz:", len(out), "arrays; z range", out["query_z"].min(), out["query_z"].max())


if __name__ == "__main__":
    only = set(sys.argv[1:])
    if not only or "json" in only:
        make_json()
    if not only or "elevation" in only:
        make_elevation()
