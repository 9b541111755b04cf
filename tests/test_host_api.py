"""CPU tests of the host side: the scenario_gym-shaped Python API, packing, the OpenSCENARIO
reader, the ABI surface of libsgym_hip.so and the no-fallback rule."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, bits_equal, load_golden, scenario_arrays


# ---------------------------------------------------------------- Trajectory (T1-T3)
def test_trajectory_normalisation_matches_reference():
    from scenario_gym_amd import Trajectory

    g = load_golden("trajectory")
    for i in range(int(g["norm/n"])):
        tr = Trajectory(g[f"norm/{i}/raw"], fields=tuple(g[f"norm/{i}/fields"]))
        assert np.array_equal(tr.data, g[f"norm/{i}/data"]), i
        assert not tr.data.flags.writeable


def test_trajectory_many_equals_the_constructor():
    """Trajectory.many (all trajectories of a scenario file normalised together, xosc.import_scenario) returns what the
    constructor returns for each of them, bit for bit: equal and different lengths, single knots, unsorted and repeated times,
    missing z / pitch / roll (whole columns or one value), missing or infinite headings (the finite-difference fill), wrapped
    headings; and it raises what the constructor raises."""
    from scenario_gym_amd.trajectory import Trajectory

    rng = np.random.default_rng(3)
    datas = []
    for _ in range(400):
        n = int(rng.choice([1, 2, 5, 40, 40, 40, 111, 111]))
        t = np.sort(rng.uniform(0, 10, n)) if rng.random() < 0.9 else rng.uniform(0, 10, n)
        d = np.column_stack([t, rng.normal(0, 50, n), rng.normal(0, 50, n), rng.normal(0, 1, n), rng.uniform(-10, 10, n),
                             rng.normal(0, 0.1, n), rng.normal(0, 0.1, n)])
        r = rng.random()
        if r < 0.15:
            d[:, 3] = np.nan
        elif r < 0.25:
            d[rng.integers(0, n), 5] = np.nan
        elif r < 0.35:
            d[:, 4] = np.nan
        elif r < 0.4:
            d[rng.integers(0, n), 4] = np.inf
        if rng.random() < 0.05 and n > 2:
            d[1, 0] = d[0, 0]
        datas.append(d)
    datas.append([[0.0, 1.0, 2.0, 0.0, 0.0, 0.0, 0.0]])  # (not an array: the constructor's own path)
    for d, x in zip(datas, Trajectory.many(datas)):
        y = Trajectory(d)
        assert x.data.shape == y.data.shape and x.data.tobytes() == y.data.tobytes() and not x.data.flags.writeable
    with pytest.raises(ValueError, match="Invalid values found for x"):
        Trajectory.many([np.array([[0, np.nan, 0, 0, 0, 0, 0], [1, 1, 1, 0, 0, 0, 0.0]])] * 5)


def test_trajectory_queries_match_reference():
    from scenario_gym_amd import Trajectory

    g = load_golden("trajectory")
    for i in range(int(g["pos/n"])):
        tr, q = Trajectory(g[f"pos/{i}/data"]), g[f"pos/{i}/q"]
        for name, ext in (("true", True), ("ff", (False, False)), ("ft", (False, True)), ("tf", (True, False))):
            assert bits_equal(np.array([tr.position_at_t(float(x), extrapolate=ext) for x in q]), g[f"pos/{i}/{name}"])
            assert bits_equal(tr.position_at_t(q, extrapolate=ext), g[f"pos/{i}/{name}"])
        none = np.array([tr.position_at_t(float(x), extrapolate=False) is None for x in q])
        assert np.array_equal(none, g[f"pos/{i}/false_is_none"])
        assert bits_equal(tr.velocity_at_t(q), g[f"pos/{i}/vel"])


def test_trajectory_known_answers():
    """The reference's own assertions: tests/test_trajectory.py:7-15, 49-128, 166-228, 255-269."""
    from scenario_gym_amd import Trajectory

    data = np.repeat(np.arange(10, dtype=np.float32)[:, None], 4, axis=1)
    traj = Trajectory(data, fields=["t", "x", "y", "h"])
    assert traj.max_t == 9 and np.allclose(traj.arclength, 9 * np.sqrt(2))
    with pytest.raises(ValueError):
        Trajectory(np.empty((3, 2)), fields=["x", "y"])
    with pytest.raises(ValueError):
        Trajectory(np.array([[0, np.nan, 0]]), fields=["t", "x", "y"])
    with pytest.raises(ValueError):
        traj.data[-1][0] = 11
    dup = Trajectory(np.array([[0, 0, 0], [1, 1, 1], [1, 5, 5], [2, 2, 2]]), fields=["t", "x", "y"])
    assert len(dup) == 3  # duplicate t rows are dropped
    one = Trajectory(np.array([[0.0, 1.0, 1.0]]), fields=["t", "x", "y"])
    assert np.allclose(one.h, 0) and np.allclose(one.position_at_t(10.0)[:2], 1) and one.max_t == 0.0
    t3 = Trajectory(np.array([[0, 0, 0], [1, 1, 1], [2, 2, 2]]), fields=["t", "x", "y"])
    assert np.allclose(t3.z, 0) and np.allclose(t3.position_at_t(0.5, extrapolate=True)[:2], 0.5)
    assert t3.position_at_t(-1.0, extrapolate=False) is None and t3.position_at_t(3.0, extrapolate=False) is None
    assert np.allclose(t3.position_at_t(-1.0, extrapolate=(False, True))[:2], 0.0)
    assert np.allclose(t3.position_at_t(3.0, extrapolate=(True, False))[:2], 2.0)
    x = np.array([-1.0, 3.0])
    assert np.allclose(t3.position_at_t(x, extrapolate=False)[:, :2], [[0, 0], [2, 2]])
    assert np.allclose(t3.position_at_t(x, extrapolate=(True, True))[:, :2], [[-1, -1], [3, 3]])
    tv = Trajectory(np.array([[0, 0, 0], [1, 0, 1], [2, 0, 2]]), fields=["t", "x", "y"])
    assert np.allclose(tv.velocity_at_t(0.5)[:2], [0, 1]) and np.allclose(tv.velocity_at_t(2.0)[:2], [0, 1])
    assert np.allclose(tv.velocity_at_t([0.5, 2.5])[:, :2], [[0, 1], [0, 0]])


# ---------------------------------------------------------------- entities / packing
def scenario_from_arrays(sc, refs):
    from scenario_gym_amd import BoundingBox, CatalogEntry, Entity, Scenario, Trajectory

    ents = []
    types = {0: "Vehicle", 1: "Pedestrian", 2: "MiscObject"}
    for i, ref in enumerate(refs):
        a, b = sc["knot_off"][i], sc["knot_off"][i + 1]
        ce = CatalogEntry(None, "x", None, types[int(sc["etype"][i])], BoundingBox(*sc["bbox"][i]))
        ents.append(Entity(ce, Trajectory(sc["knots"][a:b]), ref=str(ref)))
    return Scenario(ents)


def test_corners_match_reference():
    from scenario_gym_amd import BoundingBox, CatalogEntry, Entity

    g = load_golden("collision")
    for pose, box, ref in zip(g["corners/poses"][:200], g["corners/boxes"][:200], g["corners/points"][:200]):
        e = Entity(CatalogEntry(None, "x", None, "Vehicle", BoundingBox(*box)))
        assert np.abs(e.get_bounding_box_points(pose) - ref).max() < 1e-12


def test_pack_scenarios_matches_reference_export():
    """Scenario/Entity/Trajectory objects -> the sg_scenarios arrays, incl. the default agents split."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd.packing import pack_scenarios

    g = load_golden("scenarios")
    scs, raws = [], []
    for n in g["names"]:
        raw = scenario_arrays(g, f"{n}/scenario")
        raws.append(raw)
        scs.append(scenario_from_arrays(raw, g[f"{n}/scenario/refs"]))
    packed, agents = pack_scenarios(scs)
    E = packed.n_entities
    assert E == max(len(r["etype"]) for r in raws)
    for r, raw in enumerate(raws):
        n = len(raw["etype"])
        a, b = packed.knot_off[r * E], packed.knot_off[r * E + n]
        assert np.array_equal(packed.knots[a:b], raw["knots"])
        assert np.array_equal(packed.bbox[r * E:r * E + n], raw["bbox"])
        assert packed.t0[r] == raw["t0"] and packed.length[r] == raw["length"] and packed.ego[r] == raw["ego"]
        kind = packed.kind[r * E:(r + 1) * E]
        assert kind[raw["ego"]] == L.KIND_AGENT_REPLAY and (kind[n:] == L.KIND_NONE).all()
        assert (np.delete(kind[:n], raw["ego"]) == L.KIND_REPLAY).all()
        assert len(agents[r]) == 1
    sh = packed.shard(1, 3)
    assert sh.n_scenarios == 2 and sh.knot_off[0] == 0 and sh.t0[0] == raws[1]["t0"]


def test_packed_batches_through_shared_memory(monkeypatch):
    """Worker -> parent hand-off of packed batches (packing.packed_to_shm / merge_packed_shm / release_shm): same arrays as
    merge_packed, no segment left behind -- neither after a merge nor when the parent gives up in between --, and the
    pickle path when /dev/shm has no room."""
    import os

    import scenario_gym_amd.packing as P

    g = load_golden("scenarios")
    scs = [scenario_from_arrays(scenario_arrays(g, f"{n}/scenario"), g[f"{n}/scenario/refs"]) for n in g["names"]]
    E = max(len(s.entities) for s in scs)
    parts = [P.widen_packed(P.pack_scenarios(q)[0], E) for q in (scs[:2], scs[2:])]
    want = P.merge_packed(parts)

    def segments(metas):
        return [m["shm"] for m in metas if m["shm"] and os.path.exists("/dev/shm/" + m["shm"])]

    metas = [P.packed_to_shm(q) for q in parts]
    assert len(segments(metas)) == 2
    got = P.merge_packed_shm(metas, threads=2)
    assert segments(metas) == []
    for f in ("kind", "etype", "bbox", "knot_off", "knots", "ego", "t0", "length"):
        assert np.array_equal(getattr(got, f), getattr(want, f)), f
    one = P.packed_from_shm(P.packed_to_shm(parts[0]))
    assert np.array_equal(one.knots, parts[0].knots)

    metas = [P.packed_to_shm(q) for q in parts]      # the parent fails before the merge: it releases what it received
    assert P.release_shm(metas) == 2 and segments(metas) == [] and P.release_shm(metas) == 0
    metas = [P.packed_to_shm(q) for q in parts]      # the merge itself fails: nothing stays behind
    metas[1] = dict(metas[1], E=E + 1)
    with pytest.raises(ValueError):
        P.merge_packed_shm(metas)
    P.release_shm(metas)
    assert segments(metas) == []

    class Full:
        f_bavail, f_frsize = 0, 4096

    monkeypatch.setattr(P.os, "statvfs", lambda path: Full)
    metas = [P.packed_to_shm(q) for q in parts]
    assert all(m["shm"] is None for m in metas)
    got = P.merge_packed_shm(metas, threads=1)
    assert np.array_equal(got.knots, want.knots) and np.array_equal(got.knot_off, want.knot_off)


def test_agent_descriptors_lower_to_device_kinds():
    import scenario_gym_amd as sga
    import scenario_gym_amd._lib as L
    from scenario_gym_amd.packing import pack_scenarios

    g = load_golden("pid_xosc")
    sc = scenario_from_arrays(scenario_arrays(g, "scenario"), g["scenario/refs"])

    def create_agent(s, e):  # tests/test_controller.py:11-19
        if e.ref == "ego":
            return sga.PIDAgent(e, accel_Kp=2.0, max_accel=5.0, max_steer=np.pi / 90)

    packed, _ = pack_scenarios([sc], create_agent)
    ego = packed.ego[0]
    assert packed.kind[ego] == L.KIND_AGENT_PID
    row = packed.ctrl[ego]
    assert row[L.C_ACCEL_KP] == 2.0 and row[L.C_MAX_ACCEL] == 5.0 and row[L.C_MAX_STEER] == np.pi / 90
    assert row[L.C_STEER_KP] == 0.03054 and np.isnan(row[L.C_MAX_SPEED])

    class MyAgent(sga.Agent):  # a Python _step: runs in the caller, its slot takes injected poses
        def __init__(self, entity):
            super().__init__(entity, sga.ReplayTrajectoryController(entity), sga.EgoLocalizationSensor(entity))

        def _step(self, observation):
            return sga.TeleportAction(pose=np.zeros(6))

    packed, agents = pack_scenarios([sc], lambda s, e: MyAgent(e) if e.ref == "ego" else None)
    assert packed.kind[packed.ego[0]] == L.KIND_AGENT_EXTERNAL and isinstance(agents[0][sc.ego], MyAgent)
    a = sga.TeleportAction(pose=np.arange(6.0))
    assert np.array_equal(a.pose, np.arange(6.0))


def test_synthetic_generator_is_shardable_and_seeded():
    from scenario_gym_amd import synthetic

    a = synthetic.make_batch(192, 8, n_steps=100)
    b = synthetic.make_batch(64, 8, n_steps=100, first_scenario=64)
    sh = a.shard(64, 128)
    for k in ("kind", "bbox", "knot_off", "knots", "t0", "length"):
        assert np.array_equal(getattr(sh, k), getattr(b, k)), k
    n = np.diff(a.knot_off)
    assert (n == 1).any() and (n == 128).any() and ((n > 1) & (n < 128)).any()  # static / full / vanishing
    assert np.array_equal(synthetic.make_actions(5, 64, first_scenario=64), synthetic.make_actions(5, 128)[:, 64:])


# ---------------------------------------------------------------- OpenSCENARIO ingest
XOSC = """<?xml version="1.0"?>
<OpenSCENARIO>
  <CatalogLocations><VehicleCatalog><Directory path="cats"/></VehicleCatalog></CatalogLocations>
  <Entities>
    <ScenarioObject name="hero"><CatalogReference catalogName="C" entryName="car"/></ScenarioObject>
    <ScenarioObject name="walker"><CatalogReference catalogName="C" entryName="ped"/></ScenarioObject>
  </Entities>
  <Storyboard>
    <Init><Actions><Private entityRef="walker"><PrivateAction><TeleportAction><Position>
      <WorldPosition x="3" y="4" h="0.5"/></Position></TeleportAction></PrivateAction></Private></Actions></Init>
    <Story><Act><ManeuverGroup><Actors><EntityRef entityRef="hero"/></Actors><Maneuver><Event><Action>
      <PrivateAction><RoutingAction><FollowTrajectoryAction><Trajectory><Shape><Polyline>
        <Vertex time="0"><Position><WorldPosition x="0" y="0"/></Position></Vertex>
        <Vertex time="2"><Position><WorldPosition x="10" y="0"/></Position></Vertex>
        <Vertex time="1"><Position><WorldPosition x="5" y="1"/></Position></Vertex>
      </Polyline></Shape></Trajectory></FollowTrajectoryAction></RoutingAction></PrivateAction>
    </Action></Event></Maneuver></ManeuverGroup></Act></Story>
  </Storyboard>
</OpenSCENARIO>"""
CATALOG = """<?xml version="1.0"?>
<OpenSCENARIO><Catalog name="C">
  <Vehicle name="car" vehicleCategory="car"><BoundingBox><Center x="1.37" y="0" z="0.6"/>
    <Dimensions width="2.0" length="4.2" height="1.3"/></BoundingBox></Vehicle>
  <Pedestrian name="ped" pedestrianCategory="pedestrian"><BoundingBox><Center x="0" y="0" z="0.9"/>
    <Dimensions width="0.69" length="0.7" height="1.8"/></BoundingBox></Pedestrian>
</Catalog></OpenSCENARIO>"""


def test_xosc_reader(tmp_path):
    from scenario_gym_amd import Pedestrian, Vehicle
    from scenario_gym_amd.xosc import import_scenario

    (tmp_path / "cats").mkdir()
    (tmp_path / "cats" / "c.xosc").write_text(CATALOG)
    (tmp_path / "s.xosc").write_text(XOSC)
    s = import_scenario(str(tmp_path / "s.xosc"))
    assert [e.ref for e in s.entities] == ["ego", "pedestrian_0"] and s.ego is s.entities[0]
    ego, ped = s.entities
    assert isinstance(ego, Vehicle) and isinstance(ped, Pedestrian)
    assert (ego.bounding_box.width, ego.bounding_box.length, ego.bounding_box.center_x) == (2.0, 4.2, 1.37)
    assert np.array_equal(ego.trajectory.t, [0, 1, 2]) and ego.trajectory.data[1, 1] == 5  # sorted by time
    assert np.allclose(ego.trajectory.h[0], np.arctan2(1, 5), atol=1e-6)                    # heading filled from xy
    assert ped.is_static() and np.array_equal(ped.trajectory.data[0, :3], [0, 3, 4]) and ped.trajectory.h[0] == 0.5
    assert s.length == 2.0
    assert [e.ref for e in import_scenario(str(tmp_path / "s.xosc"), relabel=False).entities] == ["hero", "walker"]
    with pytest.raises(FileNotFoundError):
        import_scenario(str(tmp_path / "nope.xosc"))


@pytest.mark.skipif(not os.path.isdir("/root/reference/tests/input_files"), reason="build container only")
def test_xosc_reader_on_reference_inputs():
    """Same knots / boxes / refs as the reference's own import_scenario (exported in the goldens)."""
    import glob

    from scenario_gym_amd.xosc import import_scenario

    g = load_golden("scenarios")
    for n in g["names"]:
        s = import_scenario(glob.glob(f"/root/reference/tests/input_files/Scenarios/{n}*.xosc")[0])
        assert np.array_equal(np.concatenate([e.trajectory.data for e in s.entities]), g[f"{n}/scenario/knots"])
        assert [e.ref for e in s.entities] == list(g[f"{n}/scenario/refs"])


@pytest.mark.skipif(not os.path.isdir("/root/reference/tests/input_files"), reason="build container only")
def test_xosc_reader_on_all_reference_inputs():
    """All 23 OpenSCENARIO files of the reference's tests: knots, boxes, catalog types, refs, ego and length read by
    scenario_gym_amd.xosc equal what the reference's own import_scenario produced (all_scenarios.npz)."""
    from scenario_gym_amd.entity import catalog_type_code
    from scenario_gym_amd.xosc import import_scenario

    g = load_golden("all_scenarios")
    for n in g["names"]:
        s = import_scenario(f"/root/reference/tests/input_files/Scenarios/{n}.xosc")
        assert np.array_equal(np.concatenate([e.trajectory.data for e in s.entities]), g[f"{n}/scenario/knots"]), n
        assert [e.ref for e in s.entities] == list(g[f"{n}/scenario/refs"]), n
        assert np.array_equal([[e.bounding_box.width, e.bounding_box.length, e.bounding_box.center_x, e.bounding_box.center_y]
                               for e in s.entities], g[f"{n}/scenario/bbox"]), n
        assert [catalog_type_code(e) for e in s.entities] == list(g[f"{n}/scenario/etype"]), n
        assert s.entities.index(s.ego) == int(g[f"{n}/scenario/ego"]) and s.length == float(g[f"{n}/scenario/length"]), n


@pytest.mark.skipif(not os.path.isdir("/root/reference/tests/input_files/Road_Networks"), reason="build container only")
def test_road_network_reader_on_all_reference_inputs():
    """The six road-network JSON files of the reference's tests: every geometry, its rings and the unions it belongs to
    (driveable / walkable / impenetrable surface, road, intersection, lane, pavement, crossing) as the reference's own
    RoadNetwork.create_from_json composed them (roads.npz)."""
    from scenario_gym_amd.road_network import RoadNetwork

    g = load_golden("roads")
    for n in g["networks"]:
        a = RoadNetwork.create_from_json(f"/root/reference/tests/input_files/Road_Networks/{n}.json").polygon_arrays()

        def by_id(ids, ring_off, vert_off, verts, layers):
            return {str(i): (int(layers[k]), [verts[vert_off[r]:vert_off[r + 1]].tobytes() for r in range(ring_off[k], ring_off[k + 1])])
                    for k, i in enumerate(ids)}

        mine = by_id(a["ids"], a["ring_off"], a["vert_off"], a["verts"], a["layers"])
        ref = by_id(*(g[f"net/{n}/{k}"] for k in ("ids", "ring_off", "vert_off", "verts", "layers")))
        assert mine == ref, n


# ---------------------------------------------------------------- the C ABI
def test_library_exports_every_declared_symbol():
    import ctypes

    import scenario_gym_amd._lib as L

    header = open(os.path.join(ROOT, "include", "sgym.h")).read()
    declared = set(re.findall(r"^\s*(?:int|int32_t|void \*|const char \*|sg_handle \*)\s*(sg_[a-z0-9_]+)\(", header, re.M))
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)
    lib = L.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.sg_version() == L.ABI_VERSION
    assert ctypes.sizeof(L.SgConfig) == 40 and ctypes.sizeof(L.SgMetrics) == 48 and ctypes.sizeof(L.SgEvent) == 24
    from scenario_gym_amd.engine import SCEN_DTYPE

    assert SCEN_DTYPE.itemsize == 136  # sg_scenario_state
    assert ctypes.sizeof(L.SgSocialForce) == 96 and ctypes.sizeof(L.SgStateView) == 40


def test_no_cpu_fallback(monkeypatch):
    """Without a GPU the product fails loudly; it never routes through the oracle or numpy."""
    import torch

    import scenario_gym_amd as sga
    import scenario_gym_amd._lib as L

    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="sg_create failed"):
            sga.RolloutEngine(4, 4)
    pkg = os.path.join(ROOT, "scenario_gym_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert "import oracle" not in src and "from oracle" not in src and "sgym_oracle" not in src, f
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libsgym_hip.so")
    monkeypatch.setattr(L, "_lib", None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.load()


def test_terminal_condition_names():
    from scenario_gym_amd.engine import terminal_mask

    assert terminal_mask(None) == 1 and terminal_mask(["max_length", "collision", "ego_collision"]) == 7
    assert terminal_mask(["ego_off_road", "max_length"]) == 9
    with pytest.raises(ValueError):
        terminal_mask(["ego_in_the_air"])


def test_scenario_translate_and_reset_start():
    """scenario.py:157-184 / trajectory.py:287-306 (used by tests/test_state.py:218 before to_scenario)."""
    g = load_golden("scenarios")
    sc = scenario_from_arrays(scenario_arrays(g, "a5e43fe4/scenario"), g["a5e43fe4/scenario/refs"])
    t0 = sc.ego.trajectory.min_t
    assert t0 > 0
    new = sc.reset_start()
    assert new.ego.trajectory.min_t == 0.0 and sc.ego.trajectory.min_t == t0  # the original is untouched
    for a, b in zip(sc.entities, new.entities):
        assert np.array_equal(a.trajectory.data[:, 1:], b.trajectory.data[:, 1:])
        assert np.array_equal(a.trajectory.data[:, 0] - t0, b.trajectory.data[:, 0])
    shifted = sc.translate(np.array([0.0, 1.5, -2.0, 0.0, 0.0, 0.0, 0.0]))
    assert np.array_equal(shifted.ego.trajectory.data[:, 1], sc.ego.trajectory.data[:, 1] + 1.5)


def test_is_stationary():
    """trajectory.py:472-490: NaNs count as zeros."""
    from scenario_gym_amd.trajectory import is_stationary

    a = np.array([[0.0, 1.0, 2.0, np.nan, 0.3, 0.0, 0.0], [1.0, 1.0, 2.0, 0.0, 0.3, np.nan, 0.0]])
    assert is_stationary(a)
    a[1, 1] = 1.1
    assert not is_stationary(a)


@pytest.mark.skipif(not os.path.isdir("/root/reference/scenario_gym"), reason="build container only")
def test_route_finder_matches_reference():
    """pedestrian/route.py on the Greenwich network (5 pavements, 4 crossings): the connection graph, its node positions
    and breadth-first routes between random points equal the reference's RouteFinder (imported with the stand-ins; the
    centre-line interpolation it calls is the stand-in's)."""
    import subprocess
    import sys
    import json

    from scenario_gym_amd.road_network import RoadNetwork
    from scenario_gym_amd.route import RouteFinder

    path = "/root/reference/tests/input_files/Road_Networks/Greenwich_Road_Network_002.json"
    rf = RouteFinder(RoadNetwork.create_from_json(path))
    rng = np.random.default_rng(0)
    lo, hi = np.array([290.0, 288.0]), np.array([591.0, 496.0])
    queries = rng.uniform(lo, hi, (12, 2, 2))
    code = f"""
import sys, json
sys.dont_write_bytecode = True
sys.path[:0] = ["{ROOT}/tests/golden/_refstubs", "/root/reference"]
import numpy as np
from scenario_gym.road_network import RoadNetwork
from scenario_gym.pedestrian.route import RouteFinder
rf = RouteFinder(RoadNetwork.create_from_json("{path}"))
q = np.array({queries.tolist()})
routes = [rf.find_route(a, b) for a, b in q]
print(json.dumps(dict(graph={{str(k): v for k, v in rf.graph.items()}}, idx=rf.node_to_idx,
                      data={{str(k): list(map(float, v)) for k, v in rf.node_data.items()}},
                      routes=[None if r is None else r.tolist() for r in routes])))
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, MPLBACKEND="Agg"))
    assert out.returncode == 0, out.stderr[-2000:]
    ref = json.loads(out.stdout.strip().splitlines()[-1])
    assert rf.node_to_idx == ref["idx"] and len(rf.graph) > 100
    assert {str(k): v for k, v in rf.graph.items()} == ref["graph"]
    for k, v in rf.node_data.items():
        assert np.allclose(v, ref["data"][str(k)], rtol=0, atol=1e-9)
    found = 0
    for (a, b), want in zip(queries, ref["routes"]):
        got = rf.find_route(a, b)
        assert (got is None) == (want is None)
        if got is not None:
            assert np.allclose(got, np.array(want), rtol=0, atol=1e-9)
            found += 1
    assert found > 0


def test_register_budgets_of_the_two_kernel_path():
    """The table kernels of the rollout (rollout_kernel_tab<G>, _tab_planar<G>) run three wavefronts per SIMD: 168 VGPRs each
    (a wavefront of them is latency-bound, the third one is nearly free) with at most a few spilled
    registers; control_kernel stays within 128 (beside two of them).  Read from the built code object."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_resources import table

    t = table()
    for fam, spill in (("rollout_kernel_tab<", 64), ("rollout_kernel_tab_planar<", 32)):
        tabs = {n: r for n, r in t.items() if fam in n}
        assert len(tabs) == 5
        for n, r in tabs.items():
            assert r["vgpr"] + r["agpr"] <= 168, (n, r)
        assert tabs[f"void sg::{fam}64>"]["scratch"] <= spill
    ctl = t["sg::control_kernel"]
    assert ctl["vgpr"] + ctl["agpr"] <= 128 and ctl["scratch"] == 0
    # the crowd tiles against a compute unit's 160 KB of LDS ("LDS residency"): a byte too many costs a resident
    # workgroup, and nothing but the clock shows it (round 6: the riders tile + 3 KB = c5mix 15 % slower)
    CU = 160 * 1024
    lds = lambda n: t[n]["lds"]  # noqa: E731
    for v in ("crowd", "crowd_models"):
        assert 2 * lds(f"void sg::rollout_kernel_{v}<4>") <= CU
        assert 4 * lds(f"void sg::rollout_kernel_{v}<2>") <= CU
        assert 8 * lds(f"void sg::rollout_kernel_{v}<1>") <= CU
    # ... and the riders variant WITH its controller pre-pass beside it
    ctl_r = lds("sg::control_kernel_riders")
    assert 2 * lds("void sg::rollout_kernel_crowd_riders<4>") + ctl_r <= CU
    assert 4 * lds("void sg::rollout_kernel_crowd_riders<2>") + ctl_r <= CU
    assert 8 * lds("void sg::rollout_kernel_crowd_riders<1>") + ctl_r <= CU


def test_walk_graph_bfs_equals_list_of_paths_search():
    """scenario_gym_amd/route.py: the parent-pointer BFS returns the path the reference's list-of-paths search
    (pedestrian/route.py:131-158) returns -- restated here as the textbook FIFO of paths -- on random graphs whose
    adjacency order matters, including unreachable goals."""
    from types import SimpleNamespace

    from scenario_gym_amd.route import WalkGraph, find_route

    def fifo_of_paths(graph, start, goal):
        if start == goal:
            return [start]
        seen, queue = [], [[start]]
        while queue:
            path = queue.pop(0)
            node = path[-1]
            if node not in seen:
                for nb in graph[node]:
                    new = path + [nb]
                    queue.append(new)
                    if nb == goal:
                        return new
                seen.append(node)
        return None

    rng = np.random.default_rng(3)
    # a "road network" of two pavements and one crossing joining them, plus one pavement nothing leads to
    line = lambda a, b: np.array([a, b], dtype=float)
    rn = SimpleNamespace(
        pavements=[SimpleNamespace(id="p0", center=line((0, 0), (30, 0))), SimpleNamespace(id="p1", center=line((0, 12), (30, 12))),
                   SimpleNamespace(id="p2", center=line((100, 100), (108, 100)))],
        crossings=[SimpleNamespace(id="c0", center=line((15, 0.5), (15, 11.5)), pavements=["p0", "p1"])])
    walk = WalkGraph(rn)
    graph, idx, data = walk.as_dicts()
    assert len(walk) == 30 + 30 + 8 + 11 and idx["p1_0"] == 30 and idx["c0_0"] == 68
    assert graph[idx["c0_0"]][-1] == idx["p0_15"] or graph[idx["c0_0"]][-1] == idx["p0_14"]  # the closest pavement sample
    for _ in range(200):
        a, b = (int(v) for v in rng.integers(0, len(walk), 2))
        assert walk.bfs_path(a, b) == fifo_of_paths(graph, a, b), (a, b)
    assert walk.bfs_path(idx["p0_3"], idx["p2_1"]) is None
    r = find_route(walk, np.array([1.0, -1.0]), np.array([29.0, 13.0]))
    assert r is not None and np.allclose(r[0], [1, -1]) and np.allclose(r[-1], [29, 13]) and len(r) > 20
    empty = WalkGraph(SimpleNamespace(pavements=[], crossings=[]))
    assert np.array_equal(find_route(empty, np.zeros(2), np.ones(2)), np.array([[0.0, 0.0], [1.0, 1.0]]))


def test_native_xosc_scan_equals_the_elementtree_reader(tmp_path):
    """libsgym_xosc.so (include/sgym_xosc.h): exports what the header declares, and import_scenario through it returns what
    the ElementTree reader returns -- on generated directories (tools/ingest_rate.py layout) and on a hand-written file with
    the corners of the format: comments, an XML entity in a name, an inline entity definition, a TrajectoryRef, an Event
    without vertices, a second Event that replaces the first trajectory, a teleport only, attributes in single quotes."""
    import ctypes
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from ingest_rate import make_directory
    from scenario_gym_amd import xosc

    lib = ctypes.CDLL(os.path.join(ROOT, "scenario_gym_amd", "lib", "libsgym_xosc.so"))
    header = open(os.path.join(ROOT, "include", "sgym_xosc.h")).read()
    for sym in xosc.XOSC_SYMBOLS:
        assert hasattr(lib, sym) and f"int {sym}(" in header
    assert lib.sgx_version() == 1

    def same(a, b):
        assert a.name == b.name and [e.ref for e in a.entities] == [e.ref for e in b.entities]
        for x, y in zip(a.entities, b.entities):
            assert type(x) is type(y) and bits_equal(x.trajectory.data, y.trajectory.data)
            bx, by = x.catalog_entry.bounding_box, y.catalog_entry.bounding_box
            assert (bx.width, bx.length, bx.center_x, bx.center_y) == (by.width, by.length, by.center_x, by.center_y)
            assert (x.catalog_entry.catalog_type, x.catalog_entry.catalog_entry) == (y.catalog_entry.catalog_type, y.catalog_entry.catalog_entry)

    paths = make_directory(str(tmp_path / "gen"), 12, 5, 40)
    for p in paths:
        same(xosc.import_scenario(p), xosc.import_scenario_et(p))
    # the corners
    wp = lambda x, y, extra="": f'<Position><WorldPosition x="{x}" y="{y}"{extra}/></Position>'
    H = ' h="0.25"'
    vtx = lambda t, x, y: f"<Vertex time='{t}'>{wp(x, y, H)}</Vertex>"
    fta = lambda body: ("<Action name='a'><PrivateAction><RoutingAction><FollowTrajectoryAction>" + body +
                        "</FollowTrajectoryAction></RoutingAction></PrivateAction></Action>")
    traj = lambda vs: "<Trajectory name='t' closed='false'><Shape><Polyline>" + "".join(vs) + "</Polyline></Shape></Trajectory>"
    tele_a, tele_b = wp(3.5, -2.25, ' z="0" h="1.5" p="0" r="0"'), wp(0, 0)
    ev0 = fta(traj([vtx(0.0, 1.0, 2.0), vtx(1.5, 4.0, 2.5)]))
    ev2 = fta("<TrajectoryRef>" + traj([vtx(0.5, -1.0, 0.0), vtx(2.5, -3.0, 1e1), vtx(4.0, -6.5, 2.0e1)]) + "</TrajectoryRef>")
    ev3 = fta(traj([vtx(0.0, 10.0, 10.0), vtx(3.0, 11.0, 12.0)]))
    text = f"""<?xml version="1.0"?>
<!-- a comment with <Vertex time="9"> inside -->
<OpenSCENARIO><FileHeader description="x &lt; y"/>
<CatalogLocations><VehicleCatalog><Directory path="../Catalogs"/></VehicleCatalog></CatalogLocations>
<Entities>
 <ScenarioObject name="A &amp; B"><CatalogReference catalogName="SyntheticVehicleCatalog" entryName="van"/></ScenarioObject>
 <ScenarioObject name="walker"><Pedestrian name="p1" pedestrianCategory="pedestrian" mass="70" model="m">
   <BoundingBox><Center x="0.1" y="0.0" z="0.9"/><Dimensions width="0.69" length="0.7" height="1.8"/></BoundingBox></Pedestrian></ScenarioObject>
 <ScenarioObject name="parked"><CatalogReference catalogName="SyntheticVehicleCatalog" entryName="car1"/></ScenarioObject>
 <ScenarioObject name="ghost"><CatalogReference catalogName="SyntheticVehicleCatalog" entryName="no_such_entry"/></ScenarioObject>
</Entities>
<Storyboard><Init><Actions>
 <Private entityRef="parked"><PrivateAction><TeleportAction>{tele_a}</TeleportAction></PrivateAction></Private>
 <Private entityRef="A &amp; B"><PrivateAction><TeleportAction>{tele_b}</TeleportAction></PrivateAction></Private>
</Actions></Init>
<Story name="s"><Act name="a">
 <ManeuverGroup name="g0"><Actors><EntityRef entityRef="A &amp; B"/></Actors><Maneuver name="m">
  <Event name="e0">{ev0}</Event>
  <Event name="e1"><Action name="other"><UserDefinedAction/></Action></Event>
  <Event name="e2">{ev2}</Event>
 </Maneuver></ManeuverGroup>
 <ManeuverGroup name="g1"><Actors><EntityRef entityRef="walker"/></Actors><Maneuver name="m">
  <Event name="e">{ev3}</Event></Maneuver></ManeuverGroup>
</Act></Story></Storyboard></OpenSCENARIO>
"""
    p = tmp_path / "gen" / "Scenarios" / "corners.xosc"
    p.write_text(text)
    with pytest.warns(UserWarning, match="no_such_entry"):
        a = xosc.import_scenario(str(p), relabel=False)
    with pytest.warns(UserWarning):
        b = xosc.import_scenario_et(str(p), relabel=False)
    same(a, b)
    assert [e.ref for e in a.entities] == ["A & B", "walker", "parked"]
    assert a.entities[0].trajectory.data.shape == (3, 7) and a.entities[0].trajectory.data[1, 2] == 10.0   # the later Event wins
    assert type(a.entities[1]).__name__ == "Pedestrian" and a.entities[2].trajectory.data.shape == (1, 7)
    with pytest.raises(ValueError):
        xosc.scan_xosc(b"<OpenSCENARIO><Entities><ScenarioObject name=oops></Entities>")


def test_native_xosc_scan_stray_groups_numbers_and_references(tmp_path):
    """ADVICE r2 on libsgym_xosc.so: (1) a ManeuverGroup outside Storyboard/Story/Act -- placed after a valid one -- is ignored,
    it neither steals the earlier trajectory nor adds its own; (2) numbers are read as float() reads them: a value with
    trailing characters, a hex value or a Vertex without time make the scan refuse the file (the document reader then
    decides), a value longer than 64 characters is a number, not NaN; positions nobody reads are not parsed at all;
    (3) numeric character references in names resolve as a parser resolves them."""
    import numpy as np

    from scenario_gym_amd import xosc
    from scenario_gym_amd import xosc_write as W

    root = tmp_path / "gen"
    W.write_catalog(str(root / "Catalogs"))
    (root / "Scenarios").mkdir()
    wp = lambda x, y, extra="": f'<Position><WorldPosition x="{x}" y="{y}"{extra}/></Position>'
    vtx = lambda t, x, y: f'<Vertex time="{t}">{wp(x, y)}</Vertex>'
    fta = lambda vs: ("<Maneuver name='m'><Event name='e'><Action name='a'><PrivateAction><RoutingAction><FollowTrajectoryAction><Trajectory name='t' closed='false'>"
                      "<Shape><Polyline>" + "".join(vs) + "</Polyline></Shape></Trajectory></FollowTrajectoryAction></RoutingAction></PrivateAction></Action></Event></Maneuver>")
    group = lambda ref, vs: f'<ManeuverGroup name="g"><Actors><EntityRef entityRef="{ref}"/></Actors>{fta(vs)}</ManeuverGroup>'

    def doc(story, extra="", names=("a", "b")):
        objs = "".join(f'<ScenarioObject name="{n}"><CatalogReference catalogName="SyntheticVehicleCatalog" entryName="car1"/></ScenarioObject>' for n in names)
        return (f'<?xml version="1.0"?><OpenSCENARIO><CatalogLocations><VehicleCatalog><Directory path="../Catalogs"/></VehicleCatalog>'
                f'</CatalogLocations><Entities>{objs}</Entities><Storyboard><Init><Actions/></Init>{story}</Storyboard>{extra}</OpenSCENARIO>')

    def load(name, text):
        p = root / "Scenarios" / name
        p.write_text(text)
        return xosc.import_scenario(str(p), relabel=False), xosc.import_scenario_et(str(p), relabel=False)

    good = group("a", [vtx(0, 1, 2), vtx(1, 3, 4)])
    stray = group("b", [vtx(0, 9, 9), vtx(2, 8, 8)])
    # (1) stray groups: directly under Storyboard, and under a Story without an Act
    n, e = load("stray.xosc", doc(f'<Story name="s"><Act name="x">{good}</Act></Story>{stray}<Story name="t">{stray}</Story>'))
    for s in (n, e):
        assert [x.ref for x in s.entities] == ["a", "b"]
        assert s.entities[0].trajectory.data.shape == (2, 7) and s.entities[0].trajectory.data[1, 1] == 3.0
        assert s.entities[1].trajectory is None
    scan = xosc.scan_xosc((root / "Scenarios" / "stray.xosc").read_bytes())
    assert [(r, len(v)) for r, v in scan["trajectories"]] == [("a", 2)]
    # (2) numbers
    long_one = "1." + "0" * 80 + "5"
    n, e = load("long.xosc", doc(f'<Story name="s"><Act name="x">{group("a", [vtx(0, long_one, 2), vtx(1, " 3.5 ", "+4e0")])}</Act></Story>'))
    for s in (n, e):
        assert np.array_equal(s.entities[0].trajectory.data[:, 1:3], [[float(long_one), 2.0], [3.5, 4.0]])
    for bad in ("1.0abc", "0x10", "1e", "--1", ""):
        text = doc(f'<Story name="s"><Act name="x">{group("a", [vtx(0, bad, 2), vtx(1, 3, 4)])}</Act></Story>').encode()
        with pytest.raises(ValueError):
            xosc.scan_xosc(text)
    with pytest.raises(ValueError):   # a Vertex without time
        xosc.scan_xosc(doc('<Story name="s"><Act name="x">' + group("a", [f"<Vertex>{wp(1, 2)}</Vertex>"]) + "</Act></Story>").encode())
    # a malformed number in a position nobody reads does not matter (the document readers never look at it either)
    n, e = load("unused.xosc", doc(f'<Story name="s"><Act name="x">{good}</Act></Story>', extra=f"<Junk>{wp('zzz', 'q')}</Junk>"))
    assert n.entities[0].trajectory.data.shape == (2, 7) and e.entities[0].trajectory.data.shape == (2, 7)
    # inf / nan spellings float() accepts are numbers for the scan too (Trajectory rejects them afterwards, in both readers)
    scan = xosc.scan_xosc(doc(f'<Story name="s"><Act name="x">{group("a", [vtx(0, "-Infinity", "NaN"), vtx(1, 3, 4)])}</Act></Story>').encode())
    assert scan["trajectories"][0][1][0, 1] == -np.inf and np.isnan(scan["trajectories"][0][1][0, 2])
    # (3) character references
    n, e = load("refs.xosc", doc(f'<Story name="s"><Act name="x">{group("c&#65;&#x42;", [vtx(0, 1, 2), vtx(1, 3, 4)])}</Act></Story>', names=("c&#65;&#x42;", "b")))
    for s in (n, e):
        assert [x.ref for x in s.entities] == ["cAB", "b"] and s.entities[0].trajectory.data.shape == (2, 7)
