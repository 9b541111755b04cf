#!/usr/bin/env python3
"""Golden vectors for EVERY OpenSCENARIO file of the reference's tests (tests/test_scenarios.py rolls all of them out):
tests/golden/all_scenarios.npz.  Build container only, same import stand-ins as make_golden.py:

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden_all_scenarios.py

Per scenario: the numeric content (knots, boxes, catalog types, refs, ego, length) and what the REAL reference produces
with the default gym (dt = 1/30, default agents, the three ego metrics): step count, final t, the clock of every step,
final poses / velocities / distances / presence, poses of every 25th step, the metrics.  Collision adjacency of the final
state comes from the stand-in's exact-rational SAT (see _refstubs/README.md).  Only data.
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
from scenario_gym.metrics import EgoAvgSpeed, EgoDistanceTravelled, EgoMaxSpeed  # noqa: E402
from scenario_gym.xosc_interface import import_scenario  # noqa: E402

assert scenario_gym.__version__ == "0.3.1"
SCEN_DIR = "/root/reference/tests/input_files/Scenarios"
ETYPE = {"Vehicle": 0, "Pedestrian": 1}


def main():
    out, names = {}, []
    for f in sorted(os.listdir(SCEN_DIR)):
        if not f.endswith(".xosc"):
            continue
        n = os.path.splitext(f)[0]
        names.append(n)
        s = import_scenario(os.path.join(SCEN_DIR, f))
        ents = s.entities
        off = np.concatenate([[0], np.cumsum([e.trajectory.data.shape[0] for e in ents])]).astype(np.int64)
        out[f"{n}/scenario/knot_off"] = off
        out[f"{n}/scenario/knots"] = np.concatenate([e.trajectory.data for e in ents], axis=0)
        out[f"{n}/scenario/bbox"] = np.array([[e.bounding_box.width, e.bounding_box.length, e.bounding_box.center_x,
                                               e.bounding_box.center_y] for e in ents], np.float64)
        out[f"{n}/scenario/etype"] = np.array([ETYPE.get(e.catalog_entry.catalog_type, 2) for e in ents], np.int32)
        out[f"{n}/scenario/refs"] = np.array([e.ref for e in ents])
        out[f"{n}/scenario/ego"] = np.int64(ents.index(s.ego))
        out[f"{n}/scenario/length"] = np.float64(s.length)
        gym = ScenarioGym(metrics=[EgoAvgSpeed(), EgoMaxSpeed(), EgoDistanceTravelled()])
        gym.set_scenario(s)
        st = gym.state
        E = len(ents)

        def snap():
            P = np.full((E, 6), np.nan)
            for i, e in enumerate(ents):
                if e in st.poses:
                    P[i] = st.poses[e]
            return P

        ts, keyframes = [st.t], [snap()]
        k = 0
        while not st.is_done:
            gym.step()
            k += 1
            ts.append(st.t)
            if k % 25 == 0:
                keyframes.append(snap())
        V = np.full((E, 6), np.nan)
        for i, e in enumerate(ents):
            if e in st.velocities and e in st.poses:
                V[i] = st.velocities[e]
        A = np.zeros((E, E), np.uint8)
        for e, others in st.collisions().items():
            for o in others:
                A[ents.index(e), ents.index(o)] = 1
        m = gym.get_metrics()
        out[f"{n}/t"] = np.array(ts)
        out[f"{n}/keyframes"] = np.array(keyframes)
        out[f"{n}/final_poses"] = snap()
        out[f"{n}/final_vels"] = V
        out[f"{n}/final_dists"] = np.array([st.distances[e] for e in ents], np.float64)
        out[f"{n}/final_coll"] = A
        for key in ("ego_avg_speed", "ego_max_speed", "ego_distance_travelled"):
            out[f"{n}/metric_{key}"] = np.float64(m[key])
        print(n, E, k, m)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(os.environ.get("SG_GOLDEN_OUT", HERE), "all_scenarios.npz"), **out)


if __name__ == "__main__":
    main()
