#!/usr/bin/env python3
"""Golden vectors for the collision classification of CollisionMetric:
tests/golden/collision_types.npz.

Build container only (needs /root/reference and the import stand-ins of tests/golden/_refstubs, see its README):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden_collision_types.py

`CollisionMetric.record_collision` (metrics/collision.py:81-203) reads `hazard.pose` and `self.ego.pose`, attributes that
`Entity` does not have at this commit: for a Vehicle hazard the reference raises AttributeError.  The only reading that
makes the method work is "the entity's pose in the state it was handed", so this script gives Entity a `pose` property
that returns `state.poses[entity]` of the gym being rolled out and then runs the reference's own code: the classification
tree, angle_between and get_collision_point are the reference's.  The intersection polygon and the centroids come from
the stand-in (convex clip + triangle-fan area centroid), not from GEOS.  Only data is stored.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path[:0] = [HERE, os.path.join(HERE, "_refstubs"), "/root/reference"]

import numpy as np  # noqa: E402

import make_golden as MG  # noqa: E402
from scenario_gym import ScenarioGym  # noqa: E402
from scenario_gym.entity import Entity  # noqa: E402
from scenario_gym.metrics import CollisionMetric  # noqa: E402
from scenario_gym.metrics.collision import CollisionPointMetric  # noqa: E402
from scenario_gym.scenario import Scenario  # noqa: E402

CURRENT = {}
Entity.pose = property(lambda self: CURRENT["gym"].state.poses[self])
TYPES = ["other", "t_bone", "head_on", "rear_end", "side_swipe", "non_vehicle"]


def straight(p0, p1, t1, h):
    return np.array([[0.0, p0[0], p0[1], 0, h, 0, 0], [t1, p1[0], p1[1], 0, h, 0, 0]])


def main():
    rng = np.random.default_rng(31)
    out, names = {}, []
    scenes = []
    # hand-made: head-on, rear-end, t-bone, side swipe (same direction, lateral drift), oblique front corner, reversing ego
    scenes.append(("head_on", straight((-30, 0), (30, 0), 6, 0.0), [straight((30, 0.3), (-30, 0.3), 6, np.pi)]))
    scenes.append(("rear_end", straight((-30, 0), (30, 0), 6, 0.0), [straight((-10, 0.2), (10, 0.2), 6, 0.0)]))
    scenes.append(("rear_ended", straight((0, 0), (12, 0), 6, 0.0), [straight((-25, -0.2), (35, -0.2), 6, 0.0)]))
    scenes.append(("t_bone", straight((-30, 0), (30, 0), 6, 0.0), [straight((0, -30), (0, 30), 6, np.pi / 2)]))
    scenes.append(("t_boned", straight((0, -30), (0, 30), 6, np.pi / 2), [straight((-32, 2), (28, 2), 6, 0.0)]))
    scenes.append(("side_swipe", straight((-30, 0), (30, 0), 6, 0.0), [straight((-30, 3.5), (30, 0.5), 6, -0.05)]))
    scenes.append(("oblique", straight((-30, 0), (30, 0), 6, 0.0), [straight((25, 18), (-20, -14), 6, np.pi + 0.62)]))
    for k in range(40):  # random crossings of two or three hazards
        h0 = rng.uniform(-np.pi, np.pi)
        d = np.array([np.cos(h0), np.sin(h0)])
        ego = straight(-25 * d, 25 * d, 6, h0)
        haz = []
        for _ in range(int(rng.integers(1, 4))):
            h1 = rng.uniform(-np.pi, np.pi)
            e = np.array([np.cos(h1), np.sin(h1)])
            c = rng.uniform(-8, 8) * d + rng.normal(0, 1.0, 2)
            L = rng.uniform(10, 30)
            haz.append(straight(c - L * e, c + L * e, 6, h1 + rng.choice([0.0, 0.0, np.pi]) * (rng.random() < 0.2)))
        scenes.append((f"rand{k}", ego, haz))
    total = 0
    for name, ego, haz in scenes:
        ents = [MG.make_entity(ego, "ego", ctype="Vehicle")]
        for i, h in enumerate(haz):
            ents.append(MG.make_entity(h, f"entity_{i}", ctype="Vehicle" if (i + len(name)) % 5 else "Misc"))
        sc = Scenario(ents, name=name)
        gym = ScenarioGym(timestep=0.05, metrics=[CollisionMetric(), CollisionPointMetric()])
        CURRENT["gym"] = gym
        gym.set_scenario(sc)
        gym.rollout()
        ev = gym.get_metrics()["collisions"]
        pts = gym.get_metrics()["collision_points"]  # (ref, point, angle), metrics/collision.py:242-253
        assert [r for r, _, _ in pts] == [r for _, r, _ in ev]
        out.update(MG.flat(f"{name}/scenario", MG.export_scenario(sc)))
        refs = [e.ref for e in ents]
        out[f"{name}/ev_t"] = np.array([t for t, _, _ in ev], np.float64)
        out[f"{name}/ev_other"] = np.array([refs.index(r) for _, r, _ in ev], np.int64)
        out[f"{name}/ev_type"] = np.array([TYPES.index(c) for _, _, c in ev], np.int64)
        out[f"{name}/ev_point"] = np.array([[p[0], p[1], a] for _, p, a in pts], np.float64).reshape(-1, 3)
        names.append(name)
        total += len(ev)
        print(name, [(round(t, 2), r, c) for t, r, c in ev])
    out["names"] = np.array(names)
    out["types"] = np.array(TYPES)
    allt = np.concatenate([out[f"{n}/ev_type"] for n in names])
    print("events", total, {TYPES[k]: int((allt == k).sum()) for k in range(6)})
    np.savez_compressed(os.path.join(os.environ.get("SG_GOLDEN_OUT", HERE), "collision_types.npz"), **out)


if __name__ == "__main__":
    main()
