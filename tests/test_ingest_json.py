"""Scenario JSON (scenario/scenario.py:186-319) and the elevation fill at ingest
(xosc_interface/read.py:205-217, road_network/road_network.py:446-520) against goldens from the real reference
(tests/golden/make_golden_json.py)."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import bits_equal, load_golden

REF_SCEN = "/root/reference/tests/input_files/Scenarios"
REF_NETS = "/root/reference/tests/input_files/Road_Networks"


def _split(d):
    d = json.loads(json.dumps(d))
    trajs = []
    for e in d["entities"]:
        trajs.append(np.array(e["trajectory"], np.float64))
        e["trajectory"] = len(trajs) - 1
    return d, trajs


def _golden_dict(g, n, key="to_dict"):
    d = json.loads(str(g[f"{n}/{key}"]))
    for e in d["entities"]:
        e["trajectory"] = g[f"{n}/traj_{e['trajectory']}"].tolist()
    return d


def test_scenario_from_dict_to_dict_roundtrip_matches_reference():
    """The reference's to_dict of 9 shipped scenarios -> our Scenario.from_dict -> to_dict = what the reference's own
    to_json / from_json / to_dict round trip produced (entity classes by name, an unknown class name falling to Entity and
    losing its type-specific fields, catalogs, properties, files, actions, the road network by path)."""
    from scenario_gym_amd.scenario import Scenario

    g = load_golden("json")
    for n in [str(x) for x in g["names"]]:
        d = _golden_dict(g, n)
        rn = d["road_network"]
        if rn is not None:
            rn["path"] = os.path.join(REF_NETS, rn["name"] + ".json")   # (what to_json(road_network_path=<dir>) wrote)
        s = Scenario.from_dict(d)
        assert [type(e).__name__ for e in s.entities] == [str(c) for c in g[f"{n}/roundtrip_classes"]], n
        assert (s.road_network is not None) == bool(g[f"{n}/roundtrip_has_network"]), n
        got, trajs = _split(s.to_dict(road_network_path=REF_NETS))
        want = json.loads(str(g[f"{n}/roundtrip"]))
        assert got == want, n
        for i, t in enumerate(trajs):
            assert bits_equal(t, g[f"{n}/traj_{i}"]), (n, i)


def test_scenario_json_files(tmp_path):
    """to_json / from_json through a file; relative road-network paths resolve against the file's directory or
    road_network_dir; load_scenarios takes .json files."""
    from scenario_gym_amd.scenario import Scenario

    g = load_golden("json")
    n = str(g["names"][0])
    d = _golden_dict(g, n)
    (tmp_path / "Scenarios").mkdir()
    (tmp_path / "Road_Networks").mkdir()
    with open(tmp_path / "Road_Networks" / (d["road_network"]["name"] + ".json"), "w") as f:
        json.dump({"Roads": [], "Intersections": []}, f)
    s = Scenario.from_dict(dict(d, road_network=None))
    from scenario_gym_amd.road_network import RoadNetwork

    s.road_network = RoadNetwork(name=d["road_network"]["name"])
    p = str(tmp_path / "Scenarios" / "s.json")
    s.to_json(p)  # "../Road_Networks/<name>.json"
    raw = json.load(open(p))
    assert raw["road_network"] == {"path": os.path.join("../Road_Networks", d["road_network"]["name"] + ".json"), "name": d["road_network"]["name"]}
    back = Scenario.from_json(p)
    assert back.road_network is not None and back.road_network.path.endswith(d["road_network"]["name"] + ".json")
    assert [e.ref for e in back.entities] == [e.ref for e in s.entities]
    for a, b in zip(back.entities, s.entities):
        assert bits_equal(a.trajectory.data, b.trajectory.data) and a.catalog_entry.to_dict() == b.catalog_entry.to_dict()
    other = Scenario.from_json(p, road_network_dir=str(tmp_path / "Scenarios"))   # absolute directory + relative path
    assert other.road_network is not None


@pytest.mark.skipif(not os.path.isdir(REF_SCEN), reason="build container only")
def test_xosc_import_to_dict_matches_reference():
    """Our import of the reference's .xosc files (native scan and ElementTree reader) -> to_dict = the reference's: catalog
    entries with their group, mass, performance (in the reference's shifted fields), axles, properties and files; FileHeader
    properties; UserDefinedActions; the embedded road network."""
    from scenario_gym_amd.xosc import import_scenario, import_scenario_et

    g = load_golden("json")
    names = [str(x) for x in g["names"]]
    for n in names:
        want = json.loads(str(g[f"{n}/to_dict"]))
        for imp in (import_scenario, import_scenario_et):
            s = imp(os.path.join(REF_SCEN, n + ".xosc"))
            got, trajs = _split(s.to_dict())
            assert got == want, (n, imp.__name__)
            for i, t in enumerate(trajs):
                assert bits_equal(t, g[f"{n}/traj_{i}"]), (n, i)
    n = names[0]
    emb = import_scenario(os.path.join(REF_SCEN, n + ".xosc")).to_dict(road_network_path=None)["road_network"]
    assert sorted(emb.keys()) == [str(k) for k in g[f"{n}/embedded_keys"]]
    import hashlib

    def canonical(o):  # (lists the reference builds through a Python set come out in a per-process order)
        if isinstance(o, dict):
            return {k: (sorted(v) if k in ("successors", "predecessors") else canonical(v)) for k, v in o.items()}
        return [canonical(v) for v in o] if isinstance(o, list) else o

    for k, v in emb.items():
        if isinstance(v, list):
            assert len(v) == int(g[f"{n}/embedded_n/{k}"]), k
            dig = hashlib.sha256(json.dumps(sorted(canonical(v), key=lambda o: o["id"]), sort_keys=True).encode()).hexdigest()
            assert dig == str(g[f"{n}/embedded/{k}"]), k


def _elevation_case(tmp_path):
    from scenario_gym_amd import xosc_write as W

    g = load_golden("elevation")
    (tmp_path / "Road_Networks").mkdir()
    (tmp_path / "Scenarios").mkdir()
    W.write_catalog(str(tmp_path / "Catalogs"))
    with open(tmp_path / "Road_Networks" / "synthetic_elevation.json", "w") as f:
        f.write(str(g["network_json"]))
    p = str(tmp_path / "Scenarios" / "elev.xosc")
    with open(p, "w") as f:
        f.write(str(g["scenario_xosc"]))
    return g, p


def test_elevation_at_point_matches_reference():
    """RoadNetwork.elevation_at_point on a network that has elevation samples: 400 points inside and outside the samples'
    hull, the scalar and the broadcast forms -- the reference's values, bit for bit (same scipy)."""
    from scenario_gym_amd.road_network import RoadNetwork

    g = load_golden("elevation")
    rn = RoadNetwork.create_from_dict(json.loads(str(g["network_json"])))
    q = g["query_xy"]
    z = rn.elevation_at_point(q[:, 0], q[:, 1])
    assert bits_equal(z, g["query_z"])
    assert bits_equal(rn.elevation_at_point(31.5, 12.25), g["scalar_z"])
    assert bits_equal(rn.elevation_at_point(31.5, q[:5, 1]), g["broadcast_z"])
    inner = (q[:, 0] > 5) & (q[:, 0] < 55) & (q[:, 1] > 5) & (q[:, 1] < 35)   # well inside the samples: close to the surface
    assert inner.sum() > 50 and (np.abs(z - (0.05 * q[:, 0] - 0.02 * q[:, 1] + 0.001 * q[:, 0] * q[:, 1] + 2.0))[inner] < 0.3).all()
    flat = RoadNetwork.create_from_dict({"Roads": [], "Intersections": []})   # no samples: z = 0 everywhere
    assert (flat.elevation_at_point(q[:, 0], q[:, 1]) == 0.0).all()


def test_elevation_fill_at_ingest_matches_reference(tmp_path):
    """A scenario on that network whose vertices lack z: the knots both readers end up with equal the reference's (ego: all z
    from the surface; an entity with z everywhere keeps it; one with z at every other vertex has ALL of its z replaced; an
    Init-teleport-only entity is never filled)."""
    from scenario_gym_amd.xosc import import_scenario, import_scenario_et

    g, p = _elevation_case(tmp_path)
    for imp in (import_scenario, import_scenario_et):
        s = imp(p)
        assert len(s.entities) == int(g["n_entities"]) and s.road_network is not None
        for i, e in enumerate(s.entities):
            assert bits_equal(e.trajectory.data, g[f"knots_{i}"]), (imp.__name__, i)
    assert (g["knots_0"][:, 3] != 0).all() and (g["knots_1"][:, 3] == 1.25).all() and (g["knots_2"][:, 3] != -0.5).all()
    assert g["knots_3"].shape[0] == 1 and g["knots_3"][0, 3] == 0.0


@pytest.mark.gpu
def test_elevation_case_replay_bit_identical(tmp_path):
    """... and the device replay of that scenario reproduces the reference's recorded poses (z included) bit for bit."""
    import scenario_gym_amd as sga

    g, p = _elevation_case(tmp_path)
    gym = sga.ScenarioGym(timestep=0.1)
    gym.load_scenario(p)
    gym.rollout()
    rec = gym.state.recorded_poses()
    for i, e in enumerate(gym.state.scenario.entities):
        assert bits_equal(np.asarray(rec[e]), g[f"recorded_{i}"]), i


@pytest.mark.gpu
def test_load_scenarios_takes_json_and_xosc(tmp_path):
    """BatchedScenarioGym.load_scenarios on a mix of .xosc files and the .json files Scenario.to_json wrote from them: the
    two forms of a scenario roll out to the same bits."""
    import scenario_gym_amd as sga
    from scenario_gym_amd import xosc_write as W
    from scenario_gym_amd.xosc import import_scenario

    paths = list(W.make_directory(str(tmp_path), 6, 5, 30, duration=8.0, extent=20.0))
    jpaths = []
    for p in paths:
        j = p.replace(".xosc", ".json")
        import_scenario(p, relabel=True).to_json(j)
        jpaths.append(j)
    gym = sga.BatchedScenarioGym(timestep=0.1)
    gym.load_scenarios(paths + jpaths, workers=2)
    gym.rollout()
    st = gym.engine.state()
    m = gym.get_metrics()
    gym.close()
    n = len(paths)
    assert bits_equal(st["poses"][:n], st["poses"][n:]) and bits_equal(st["dists"][:n], st["dists"][n:])
    assert m[:n] == m[n:]


def test_bulk_ingest_equals_object_path(tmp_path, monkeypatch):
    """packing.load_and_pack: OpenSCENARIO files on the usual path are packed straight from the native scan (no Scenario /
    Entity / Trajectory / Agent objects); SG_INGEST_OBJECTS=1 sends every file through import_scenario + pack_scenarios.
    Same arrays, bit for bit -- generated files (several widths in one batch, a trajectory given out of time order and one
    without headings, so that the constructor's slow branches run) and, in the build container, every shipped scenario of
    the reference (road networks, elevation, catalogs of three kinds, files the strict scan hands to the document reader)."""
    import glob

    from scenario_gym_amd import xosc_write as W
    from scenario_gym_amd.packing import load_and_pack

    root = str(tmp_path)
    W.write_catalog(os.path.join(root, "Catalogs"))
    os.makedirs(os.path.join(root, "Scenarios"))
    paths = []
    for i, (E, V) in enumerate([(6, 30), (40, 111), (3, 5), (12, 64), (1, 9)]):
        rng = np.random.default_rng([7, i])
        ents = W.synthetic_entities(rng, E, V, duration=8.0, extent=40.0)
        p = os.path.join(root, "Scenarios", f"s{i}.xosc")
        W.write_scenario(p, ents)
        paths.append(p)
    sets = [paths]
    ref = sorted(glob.glob("/root/reference/tests/input_files/Scenarios/*.xosc"))
    if ref:
        sets.append(ref)
    for ps in sets:
        for relabel in (True, False):
            monkeypatch.delenv("SG_INGEST_OBJECTS", raising=False)
            a = load_and_pack(ps, relabel=relabel)
            monkeypatch.setenv("SG_INGEST_OBJECTS", "1")
            b = load_and_pack(ps, relabel=relabel)
            assert (a.n_scenarios, a.n_entities) == (b.n_scenarios, b.n_entities)
            for k in ("kind", "etype", "knot_off", "ego"):
                assert np.array_equal(getattr(a, k), getattr(b, k)), (relabel, k)
            for k in ("bbox", "t0", "length", "ctrl", "knots"):
                assert bits_equal(getattr(a, k), getattr(b, k)), (relabel, k)
            assert a.refs == b.refs, relabel
    monkeypatch.delenv("SG_INGEST_OBJECTS", raising=False)
