#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/*.npz from the REAL reference.

Runs only in the build container (needs /root/reference).  Usage:

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden.py

The reference is imported from /root/reference with the import stand-ins of
tests/golden/_refstubs (see its README: lxml/shapely/... are not installable here).
Everything recorded below about trajectories, batch replay, state bookkeeping,
controllers and ego metrics is computed by the reference's own numpy/scipy code.
Collision adjacency comes from the stand-in's exact-rational SAT (not GEOS).

Only DATA is written: numeric scenario content (knots, boxes, kinds) and the
reference's outputs for it.  No reference source text is stored.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path[:0] = [os.path.join(HERE, "_refstubs"), "/root/reference"]

import math  # noqa: E402
from fractions import Fraction  # noqa: E402

import numpy as np  # noqa: E402

import scenario_gym  # noqa: E402
from scenario_gym import ScenarioGym  # noqa: E402
from scenario_gym.action import VehicleAction  # noqa: E402
from scenario_gym.agent import Agent, PIDAgent, _create_agent  # noqa: E402
from scenario_gym.catalog_entry import BoundingBox, CatalogEntry  # noqa: E402
from scenario_gym.controller import VehicleController  # noqa: E402
from scenario_gym.entity import Entity  # noqa: E402
from scenario_gym.entity.batch import BatchReplayEntity  # noqa: E402
from scenario_gym.metrics import (  # noqa: E402
    CollisionMetric,
    EgoAvgSpeed,
    EgoDistanceTravelled,
    EgoMaxSpeed,
)
from scenario_gym.scenario import Scenario  # noqa: E402
from scenario_gym.sensor import EgoLocalizationSensor  # noqa: E402
from scenario_gym.trajectory import Trajectory  # noqa: E402
from scenario_gym.xosc_interface import import_scenario  # noqa: E402

assert scenario_gym.__version__ == "0.3.1"
ONLY = set(sys.argv[1:])  # e.g. `make_golden.py pedestrian`
SCEN_DIR = "/root/reference/tests/input_files/Scenarios"

ETYPE = {"Vehicle": 0, "Pedestrian": 1}


# --------------------------------------------------------------------------- helpers
def export_scenario(s):
    """Numeric content of a reference Scenario (no text, no XML)."""
    ents = s.entities
    off = [0]
    for e in ents:
        off.append(off[-1] + e.trajectory.data.shape[0])
    return dict(
        n_entities=np.int64(len(ents)),
        knot_off=np.array(off, np.int64),
        knots=np.concatenate([e.trajectory.data for e in ents], axis=0),
        bbox=np.array(
            [
                [
                    e.bounding_box.width,
                    e.bounding_box.length,
                    e.bounding_box.center_x,
                    e.bounding_box.center_y,
                ]
                for e in ents
            ],
            np.float64,
        ),
        etype=np.array([ETYPE.get(e.catalog_entry.catalog_type, 2) for e in ents], np.int32),
        refs=np.array([e.ref for e in ents]),
        ego=np.int64(ents.index(s.ego)),
        length=np.float64(s.length),
    )


def record_rollout(gym, max_steps=100000, extra=None):
    """rollout() of scenario_gym/scenario_gym.py:256-267 with per-step recording."""
    st = gym.state
    ents = st.scenario.entities
    E = len(ents)
    ts, poses, vels, dists, coll, extras = [], [], [], [], [], []

    def snap():
        ts.append(st.t)
        P = np.full((E, 6), np.nan)
        V = np.full((E, 6), np.nan)
        for i, e in enumerate(ents):
            if e in st.poses:
                P[i] = st.poses[e]
            if e in st.velocities:
                V[i] = st.velocities[e]
        poses.append(P)
        vels.append(V)
        dists.append([st.distances[e] for e in ents])
        A = np.zeros((E, E), np.uint8)
        for e, others in st.collisions().items():
            for o in others:
                A[ents.index(e), ents.index(o)] = 1
        coll.append(A)
        if extra is not None:
            extras.append(extra(gym))

    gym.reset_scenario()
    snap()
    n = 0
    while not st.is_done and n < max_steps:
        gym.step()
        snap()
        n += 1
    out = dict(
        t=np.array(ts),
        poses=np.array(poses),
        vels=np.array(vels),
        dists=np.array(dists, np.float64),
        coll=np.array(coll),
        n_steps=np.int64(n),
        is_done=np.bool_(st.is_done),
    )
    if extra is not None:
        out["extra"] = np.array(extras, np.float64)
    m = gym.get_metrics()
    for k in ("ego_avg_speed", "ego_max_speed", "ego_distance_travelled"):
        if k in m:
            out["metric_" + k] = np.float64(m[k])
    if "collisions" in m:
        ev = m["collisions"]
        out["ev_t"] = np.array([e[0] for e in ev], np.float64)
        out["ev_other"] = np.array([[x.ref for x in ents].index(e[1]) for e in ev], np.int64)
        out["ev_type"] = np.array([e[2] for e in ev])
    return out


def std_metrics():
    return [EgoAvgSpeed(), EgoMaxSpeed(), EgoDistanceTravelled(), CollisionMetric()]


def flat(prefix, d):
    return {f"{prefix}/{k}": v for k, v in d.items()}


def make_entity(data, ref, bbox=(2.0, 4.2, 1.37, 0.0), ctype="Misc"):
    ce = CatalogEntry(None, "e", "e", ctype, BoundingBox(*bbox), {}, [])
    return Entity(ce, trajectory=Trajectory(np.asarray(data, np.float64)), ref=ref)


class ExternalActionAgent(Agent):
    """Feeds a preset (accel, steer) sequence to the reference VehicleController.

    This is the reference's plugin API (Agent._step -> Action) playing the role of
    the external action of integrations/openaigym.py:197-204.
    """

    def __init__(self, entity, actions, **kw):
        super().__init__(entity, VehicleController(entity, **kw), EgoLocalizationSensor(entity))
        self.actions = actions
        self.k = 0

    def _reset(self):
        self.k = 0

    def _step(self, obs):
        a = self.actions[self.k]
        self.k += 1
        return VehicleAction(a[0], a[1])


# --------------------------------------------------------------------------- synthetic mini scenes
def synth_scene(rng, E, K, L, crowd=20.0, static_frac=0.2, vanish_frac=0.25, ragged=True):
    """Small random scene: arcs on (optionally) per-entity knot grids."""
    ents = []
    for i in range(E):
        ref = "ego" if i == 0 else f"entity_{i}"
        u = rng.random()
        if i > 0 and u < static_frac:
            data = np.array([[rng.uniform(0, L), *rng.uniform(-crowd, crowd, 2), 0.0, rng.uniform(-3, 3), 0, 0]])
            ents.append(make_entity(data, ref))
            continue
        k = K if (i == 0 or not ragged) else int(rng.integers(2, K + 1))
        if i > 0 and u < static_frac + vanish_frac:
            a, b = np.sort(rng.uniform(0.05 * L, 0.95 * L, 2))
            t = np.linspace(a, b, k)
        elif ragged and i > 0:
            t = np.sort(np.concatenate([[0.0, L], rng.uniform(0, L, k - 2)])) if k > 2 else np.array([0.0, L])
        else:
            t = np.linspace(0.0, L, k)
        x0, y0 = rng.uniform(-crowd, crowd, 2)
        h0 = rng.uniform(-math.pi, math.pi)
        v = rng.uniform(2, 12)
        kappa = rng.normal(0, 0.02)
        h = h0 + kappa * v * (t - t[0])
        dt = np.diff(t, prepend=t[0])
        x = x0 + np.cumsum(v * np.cos(h) * dt)
        y = y0 + np.cumsum(v * np.sin(h) * dt)
        data = np.stack([t, x, y, np.zeros_like(t), h, np.zeros_like(t), np.zeros_like(t)], axis=1)
        ents.append(make_entity(data, ref))
    return Scenario(ents, name="synthetic")


# --------------------------------------------------------------------------- groups
def g_trajectory(rng):
    out = {}
    # normalisation (Trajectory.__init__) incl. heading fill + unwrap + dedup
    cases = []
    for n in (1, 2, 5, 40):
        t = np.sort(rng.uniform(0, 20, n))
        raw = np.stack([t, rng.normal(0, 30, n), rng.normal(0, 30, n)], axis=1)
        cases.append((raw, ("t", "x", "y")))
        rawh = np.concatenate([raw, rng.uniform(-10, 10, (n, 1))], axis=1)
        cases.append((rawh, ("t", "x", "y", "h")))
    full = np.stack([np.array([3.0, 1.0, 2.0, 1.0, 0.5])] + [rng.normal(0, 5, 5) for _ in range(6)], axis=1)
    full[1, 3] = np.nan  # z with a NaN -> whole column zeroed
    cases.append((full, ("t", "x", "y", "z", "h", "p", "r")))
    for i, (raw, fields) in enumerate(cases):
        tr = Trajectory(raw, fields=fields)
        out[f"norm/{i}/raw"] = raw
        out[f"norm/{i}/fields"] = np.array(fields)
        out[f"norm/{i}/data"] = tr.data
    out["norm/n"] = np.int64(len(cases))
    # position_at_t / velocity_at_t
    for i, n in enumerate((1, 2, 7, 64)):
        t = np.sort(rng.uniform(0, 20, n))
        data = np.concatenate([t[:, None], rng.normal(0, 30, (n, 6))], axis=1)
        tr = Trajectory(data)
        q = np.concatenate([rng.uniform(-5, 25, 60), tr.data[:, 0], [tr.min_t - 1e-9, tr.max_t + 1e-9]])
        out[f"pos/{i}/data"] = tr.data
        out[f"pos/{i}/q"] = q
        for name, ext in (("true", True), ("ff", (False, False)), ("ft", (False, True)), ("tf", (True, False))):
            out[f"pos/{i}/{name}"] = np.array([tr.position_at_t(float(x), extrapolate=ext) for x in q])
        none = np.array([tr.position_at_t(float(x), extrapolate=False) is None for x in q])
        out[f"pos/{i}/false_is_none"] = none
        out[f"pos/{i}/false"] = np.array(
            [np.full(6, np.nan) if m else tr.position_at_t(float(x), extrapolate=False) for x, m in zip(q, none)]
        )
        out[f"pos/{i}/vel"] = np.array([tr.velocity_at_t(float(x)) for x in q])
    out["pos/n"] = np.int64(4)
    return out


def g_batch(rng):
    out = {}

    class FakeState:
        next_t = 0.0

    def run(ents, q, persist, key):
        b = BatchReplayEntity(persist=persist)
        b.add_entities(ents, [e.trajectory for e in ents])
        P = np.full((len(q), len(ents), 6), np.nan)
        for i, t in enumerate(q):
            FakeState.next_t = float(t)
            for e, p in b.step(FakeState).items():
                P[i, ents.index(e)] = p
        out[f"{key}/q"] = np.asarray(q, np.float64)
        out[f"{key}/poses"] = P
        off = [0]
        for e in ents:
            off.append(off[-1] + len(e.trajectory))
        out[f"{key}/knot_off"] = np.array(off, np.int64)
        out[f"{key}/knots"] = np.concatenate([e.trajectory.data for e in ents])

    # tests/test_entity.py:40-77 case
    e1 = make_entity(np.array([[0.0, 0, 0, 0, 0, 0, 0]]), "a")
    e2 = make_entity(np.array([[0.0, 1, 0, 0, 0, 0, 0], [2.0, 2, 0, 0, 0, 0, 0]]), "b")
    run([e1, e2], [1.0, 5.0], False, "kat")
    for c, persist in enumerate((False, True)):
        ents = []
        for i, n in enumerate((1, 2, 9, 33)):
            t = np.sort(rng.uniform(1, 15, n))
            ents.append(make_entity(np.concatenate([t[:, None], rng.normal(0, 20, (n, 6))], axis=1), f"e{i}"))
        allk = np.concatenate([e.trajectory.data[:, 0] for e in ents])
        q = np.concatenate([rng.uniform(-2, 18, 180), allk[:18], [allk.min(), allk.max()]])
        run(ents, q, persist, f"rand{c}")
    return out


def g_scenarios():
    out = {}
    names = [
        "a5e43fe4-646a-49ba-82ce-5f0063776566",
        "3fee6507-fd24-432f-b781-ca5676c834ef",
        "41dac6fa-6f83-461e-a145-08692da5f3c7",
        "5c5188e0-715a-4dd2-a6b2-b3c96b52d608",
        "a98d5c7d-76aa-49bf-b88c-97db5d5c7433",
    ]
    out["names"] = np.array([n[:8] for n in names])
    for n in names:
        s = import_scenario(os.path.join(SCEN_DIR, n + ".xosc"))
        key = n[:8]
        out.update(flat(f"{key}/scenario", export_scenario(s)))
        for dt_name, dt in (("dt30", 1.0 / 30.0), ("dt10", 0.1)):
            # Vehicle hazards hit the reference's AttributeError (collision.py:94), so
            # CollisionMetric is only attached where the ego never collides.
            gym = ScenarioGym(timestep=dt, metrics=[EgoAvgSpeed(), EgoMaxSpeed(), EgoDistanceTravelled()])
            gym.set_scenario(s)
            out.update(flat(f"{key}/{dt_name}", record_rollout(gym)))
    # vanishing variant (tests/test_scenario_gym.py:17-25), persist False/True at dt=0.1
    s = import_scenario(os.path.join(SCEN_DIR, names[0] + ".xosc"))
    s = s.copy()
    d = s.entities[1].trajectory.data.copy()
    s.entities[1].trajectory = Trajectory(d[np.logical_and(d[:, 0] < 16.5, d[:, 0] > 2.0)])
    out.update(flat("vanish/scenario", export_scenario(s)))
    for pname, persist in (("nopersist", False), ("persist", True)):
        gym = ScenarioGym(timestep=0.1, persist=persist, metrics=[EgoAvgSpeed(), EgoMaxSpeed(), EgoDistanceTravelled()])
        gym.set_scenario(s)
        out.update(flat(f"vanish/{pname}", record_rollout(gym)))
    return out


def g_synth(rng):
    """Small synthetic scenes through the reference: replay, PID ego, external actions."""
    out = {}
    L = 12.0
    n_scenes = 4
    out["n"] = np.int64(n_scenes)
    for i in range(n_scenes):
        s = synth_scene(rng, E=8 if i < 3 else 16, K=12, L=L, ragged=(i % 2 == 0))
        out.update(flat(f"{i}/scenario", export_scenario(s)))
        for dt_name, dt in (("dt30", 1.0 / 30.0), ("dt10", 0.1)):
            for pname, persist in (("nopersist", False), ("persist", True)):
                gym = ScenarioGym(timestep=dt, persist=persist, metrics=std_metrics())
                gym.set_scenario(s)
                out.update(flat(f"{i}/replay_{dt_name}_{pname}", record_rollout(gym)))
            # terminal conditions: collision / ego_collision
            for term in ("collision", "ego_collision"):
                gym = ScenarioGym(timestep=dt, terminal_conditions=["max_length", term], metrics=std_metrics())
                gym.set_scenario(s)
                out.update(flat(f"{i}/term_{term}_{dt_name}", record_rollout(gym)))

            # PID ego, default gains (controller.py:157-161)
            def pid_agent(sc, e):
                if e.ref == "ego":
                    return PIDAgent(e)

            def pid_extra(g):
                c = g.state.agents[g.state.scenario.ego].controller
                return [c.speed, c.e_lon_prev, c.e_lat_prev, c.e_lon_int]

            gym = ScenarioGym(timestep=dt, metrics=std_metrics())
            gym.set_scenario(s, create_agent=pid_agent)
            out.update(flat(f"{i}/pid_{dt_name}", record_rollout(gym, extra=pid_extra)))

            # external (accel, steer) actions on a VehicleController ego
            T = int(L / dt) + 8
            acts = np.stack([rng.uniform(-6, 6, T), rng.uniform(-0.9, 0.9, T)], axis=1)

            def ext_agent(sc, e, acts=acts):
                if e.ref == "ego":
                    return ExternalActionAgent(e, acts)

            def ext_extra(g):
                return [g.state.agents[g.state.scenario.ego].controller.speed]

            gym = ScenarioGym(timestep=dt, metrics=std_metrics())
            gym.set_scenario(s, create_agent=ext_agent)
            r = record_rollout(gym, extra=ext_extra)
            r["actions"] = acts
            out.update(flat(f"{i}/ext_{dt_name}", r))
    return out


def g_pid_xosc():
    """tests/test_controller.py:7-25 configuration."""
    s = import_scenario(os.path.join(SCEN_DIR, "a98d5c7d-76aa-49bf-b88c-97db5d5c7433.xosc"))

    def create_agent(sc, e):
        if e.ref == "ego":
            return PIDAgent(e, accel_Kp=2.0, max_accel=5.0, max_steer=np.pi / 90)

    def extra(g):
        c = g.state.agents[g.state.scenario.ego].controller
        return [c.speed, c.e_lon_prev, c.e_lat_prev, c.e_lon_int]

    gym = ScenarioGym(timestep=0.1, metrics=[EgoAvgSpeed(), EgoMaxSpeed(), EgoDistanceTravelled()])
    gym.set_scenario(s, create_agent=create_agent)
    out = flat("scenario", export_scenario(s))
    out.update(flat("run", record_rollout(gym, extra=extra)))
    out["params"] = np.array([2.0, 5.0, np.pi / 90])
    return out


def g_collision(rng):
    out = {}
    # tests/test_utils.py:12-61 scene at dt=0.1 with CollisionMetric
    box = BoundingBox(2.0, 5.0, 0.0, 0.0)
    ce = CatalogEntry("car", "car", "car", "car", box, {}, [])
    ego, haz = Entity(ce, ref="ego"), Entity(ce, ref="entity_1")
    ego.trajectory = Trajectory(np.array([[0.0, 0, 0], [10, 20, 0]]), fields=["t", "x", "y"])
    haz.trajectory = Trajectory(np.array([[0.0, 40, 0], [10, 20, 0]]), fields=["t", "x", "y"])
    s = Scenario([ego, haz])
    gym = ScenarioGym(timestep=0.1, metrics=std_metrics())
    gym.set_scenario(s)
    out.update(flat("headon/scenario", export_scenario(s)))
    out.update(flat("headon/run", record_rollout(gym)))

    # corners from Entity.get_bounding_box_points for random poses / boxes
    n = 1000
    poses = np.concatenate([rng.uniform(-200, 200, (n, 3)), rng.uniform(-40, 40, (n, 1)), rng.normal(0, 1, (n, 2))], axis=1)
    boxes = np.stack([rng.uniform(0.3, 3, n), rng.uniform(0.3, 12, n), rng.uniform(-2, 2, n), rng.uniform(-1, 1, n)], axis=1)
    corners = np.empty((n, 4, 2))
    for i in range(n):
        e = make_entity(np.array([[0.0, 0, 0, 0, 0, 0, 0]]), "x", bbox=tuple(boxes[i]))
        corners[i] = e.get_bounding_box_points(poses[i])
    out["corners/poses"] = poses
    out["corners/boxes"] = boxes
    out["corners/points"] = corners

    # random OBB pairs labelled by exact rational SAT on the reference's fp64 corners
    from shapely.geometry.base import convex_intersects_exact

    m = 10000
    A = np.empty((m, 4, 2))
    B = np.empty((m, 4, 2))
    lab = np.empty(m, np.uint8)
    pa = np.empty((m, 6))
    pb = np.empty((m, 6))
    ba = np.empty((m, 4))
    bb = np.empty((m, 4))
    for i in range(m):
        ba[i] = [rng.uniform(0.5, 2.5), rng.uniform(0.5, 6), rng.uniform(-1.5, 1.5), rng.uniform(-0.5, 0.5)]
        bb[i] = [rng.uniform(0.5, 2.5), rng.uniform(0.5, 6), rng.uniform(-1.5, 1.5), rng.uniform(-0.5, 0.5)]
        pa[i] = [*rng.uniform(-5, 5, 2), 0, rng.uniform(-7, 7), 0, 0]
        sep = rng.uniform(0, 7)
        ang = rng.uniform(0, 2 * math.pi)
        pb[i] = [pa[i, 0] + sep * math.cos(ang), pa[i, 1] + sep * math.sin(ang), 0, rng.uniform(-7, 7), 0, 0]
        if i % 10 == 0:  # axis-aligned, frequently touching along a grid
            pa[i, 3] = 0.0
            pb[i, 3] = math.pi / 2 * rng.integers(0, 4)
        ea = make_entity(np.zeros((1, 7)), "a", bbox=tuple(ba[i]))
        eb = make_entity(np.zeros((1, 7)), "b", bbox=tuple(bb[i]))
        A[i] = ea.get_bounding_box_points(pa[i])
        B[i] = eb.get_bounding_box_points(pb[i])
        lab[i] = convex_intersects_exact([tuple(p) for p in A[i]], [tuple(p) for p in B[i]])
    out["pairs/pose_a"], out["pairs/pose_b"] = pa, pb
    out["pairs/box_a"], out["pairs/box_b"] = ba, bb
    out["pairs/corners_a"], out["pairs/corners_b"] = A, B
    out["pairs/intersects"] = lab
    return out


def g_pedestrian(rng):
    """Social force (pedestrian/social_force.py), PedestrianAgent goal update and PedestrianController.

    Noise is switched off (std_lon = std_lat = 0: np.random.normal(b, 0) returns b exactly), the road
    network is an empty RoadNetwork (no boundary forces, pedestrian/social_force.py:86-104) and the
    shapely stand-ins answer LineString.project analytically and Point.buffer/contains with the exact
    64-gon rule, so every number below is the reference's own arithmetic."""
    from types import SimpleNamespace as NS

    from scenario_gym.entity import Pedestrian
    from scenario_gym.pedestrian.agent import PedestrianAgent
    from scenario_gym.pedestrian.controller import PedestrianController
    from scenario_gym.pedestrian.social_force import SocialForce, SocialForceParameters
    from scenario_gym.road_network import RoadNetwork
    from shapely.geometry import MultiPolygon

    out = {}
    params = SocialForceParameters(std_lon=0.0, std_lat=0.0)
    sf = SocialForce(params)
    out["params"] = np.array([params.relaxation_time, params.ped_repulse_V, params.ped_repulse_sigma,
                              params.ped_attract_C, params.sight_weight, float(params.sight_weight_use),
                              params.sight_angle, params.max_speed_factor, params.bias_lon, params.bias_lat])
    # ---- (a) one behaviour step on random observations with 0..6 neighbours ----
    n = 600
    poses = np.concatenate([rng.uniform(-20, 20, (n, 2)), np.zeros((n, 1)), rng.uniform(-3, 3, (n, 1)), np.zeros((n, 2))], 1)
    vels = np.concatenate([rng.normal(0, 1.2, (n, 2)), np.zeros((n, 4))], 1)
    goals = poses[:, :2] + rng.normal(0, 8, (n, 2))
    vdes = rng.uniform(0.5, 2.0, n)
    head = np.where(rng.random(n) < 0.5, 0.0, rng.uniform(-0.6, 0.6, n))
    dts = np.where(rng.random(n) < 0.5, 1 / 30, 0.1)
    nn = rng.integers(0, 7, n)
    nb_pose = np.full((n, 6, 6), np.nan)
    nb_vel = np.full((n, 6, 6), np.nan)
    res = np.empty((n, 4))
    for i in range(n):
        near = []
        for k in range(nn[i]):
            pp = np.zeros(6)
            pp[:2] = poses[i, :2] + rng.normal(0, 1.5, 2)
            vv = np.zeros(6)
            vv[:2] = rng.normal(0, 1.2, 2) if k % 3 else 0.0  # some neighbours stand still
            nb_pose[i, k], nb_vel[i, k] = pp, vv
            near.append((None, pp, vv))
        obs = NS(pose=poses[i], velocity=vels[i], t=1.0, next_t=1.0 + dts[i], head_rot_angle=head[i], near_peds=near,
                 walkable_surface=MultiPolygon(), impenetrable_surface=MultiPolygon())
        agent = NS(route=[goals[i]], goal_idx=0, speed_desired=vdes[i], force=None)
        speed, heading = sf._step(obs, agent)
        res[i] = [speed, heading, agent.force[0], agent.force[1]]
    out.update({"step/pose": poses, "step/vel": vels, "step/goal": goals, "step/vdes": vdes, "step/head": head,
                "step/dt": dts, "step/n": nn, "step/nb_pose": nb_pose, "step/nb_vel": nb_vel, "step/out": res})

    # ---- (b) closed loop: 12 pedestrians + 1 replayed vehicle, empty road network ----
    def scene(seed, n_ped, side, n_wp):
        r = np.random.default_rng(seed)
        ents = [make_entity(np.array([[0.0, -side, 0.3, 0, 0.0, 0, 0], [12.0, side, 0.5, 0, 0.0, 0, 0]]), "ego", ctype="Vehicle")]
        routes, vds = {}, {}
        for i in range(n_ped):
            start = r.uniform(-side, side, 2)
            ce = CatalogEntry(None, "p", "p", "Pedestrian", BoundingBox(0.69, 0.7, 0.0, 0.0), {}, [])
            t_end = 12.0
            e = Pedestrian(ce, Trajectory(np.array([[0.0, *start, 0, r.uniform(-3, 3), 0, 0],
                                                    [t_end, *(start + r.normal(0, 0.5, 2)), 0, 0.0, 0, 0]])), ref=f"ped_{i}")
            wps = [start + r.normal(0, 0.05, 2)]
            for _ in range(n_wp - 1):
                wps.append(-wps[-1] * r.uniform(0.3, 1.0) + r.normal(0, 1.0, 2))
            routes[e.ref] = np.array(wps)
            vds[e.ref] = r.uniform(0.5, 1.5) * 1.3
            ents.append(e)
        return Scenario(ents, name="crowd", road_network=RoadNetwork(roads=[], intersections=[])), routes, vds

    for si, (seed, n_ped, side, n_wp, thr) in enumerate([(1, 12, 3.0, 3, 3.0), (2, 20, 4.0, 2, 1.0)]):
        sc, routes, vds = scene(seed, n_ped, side, n_wp)
        out.update(flat(f"loop{si}/scenario", export_scenario(sc)))
        refs = [e.ref for e in sc.entities]
        R = np.full((len(refs), n_wp, 2), np.nan)
        for k, ref in enumerate(refs):
            if ref in routes:
                R[k] = routes[ref]
        out[f"loop{si}/routes"] = R
        out[f"loop{si}/vdes"] = np.array([vds.get(ref, np.nan) for ref in refs])
        out[f"loop{si}/distance_threshold"] = np.float64(thr)
        for dt_name, dt in (("dt30", 1.0 / 30.0), ("dt10", 0.1)):
            def create_agent(s, e, routes=routes, vds=vds, thr=thr):
                if e.ref == "ego":
                    return _create_agent(s, e)
                return PedestrianAgent(e, routes[e.ref], vds[e.ref], SocialForce(SocialForceParameters(std_lon=0.0, std_lat=0.0)),
                                       distance_threshold=thr)

            def extra(g):
                rows = []
                for e in g.state.scenario.entities:
                    a = g.state.agents.get(e)
                    if isinstance(a, PedestrianAgent):
                        rows.append([a.controller.speed, float(a.goal_idx), a.force[0], a.force[1]])
                    else:
                        rows.append([np.nan] * 4)
                return rows

            gym = ScenarioGym(timestep=dt, metrics=std_metrics())
            gym.set_scenario(sc, create_agent=create_agent)
            out.update(flat(f"loop{si}/{dt_name}", record_rollout(gym, extra=extra)))

    # ---- (c) PedestrianController._step in isolation ----
    m = 300
    cp = np.concatenate([rng.uniform(-50, 50, (m, 3)), rng.uniform(-3, 3, (m, 3))], 1)
    act = np.stack([rng.uniform(-7, 7, m), rng.uniform(-7, 7, m)], 1)
    cdt = rng.uniform(0.01, 0.2, m)
    cout = np.empty((m, 7))
    ent = make_entity(np.zeros((1, 7)), "p", ctype="Pedestrian")
    for i in range(m):
        ctl = PedestrianController(ent, max_speed=5.0)
        st = NS(poses={ent: cp[i]}, dt=cdt[i])
        cout[i, :6] = ctl._step(st, NS(speed=act[i, 0], heading=act[i, 1]))
        cout[i, 6] = ctl.speed
    out.update({"ctrl/pose": cp, "ctrl/action": act, "ctrl/dt": cdt, "ctrl/out": cout})
    return out


def main():
    rng = np.random.default_rng(20240807)
    if ONLY:  # regenerate a single group without touching the others
        groups = {"pedestrian": g_pedestrian(np.random.default_rng(20240808))} if "pedestrian" in ONLY else {}
        for name, d in groups.items():
            path = os.path.join(os.environ.get("SG_GOLDEN_OUT", HERE), name + ".npz")
            np.savez_compressed(path, **d)
            print(f"{name}: {len(d)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")
        return
    groups = dict(
        trajectory=g_trajectory(rng),
        batch=g_batch(rng),
        scenarios=g_scenarios(),
        synth=g_synth(rng),
        pid_xosc=g_pid_xosc(),
        collision=g_collision(rng),
    )
    groups["pedestrian"] = g_pedestrian(np.random.default_rng(20240808))
    for name, d in groups.items():
        path = os.path.join(os.environ.get("SG_GOLDEN_OUT", HERE), name + ".npz")
        np.savez_compressed(path, **d)
        print(f"{name}: {len(d)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
