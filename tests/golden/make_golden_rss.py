#!/usr/bin/env python3
"""Golden vectors for the RSS distances callback and metric: tests/golden/rss.npz.

Build container only (needs /root/reference and the import stand-ins of tests/golden/_refstubs, see its README):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden_rss.py

ScenarioGym(state_callbacks=[RSSDistances()], metrics=[RSS()]) (tests/test_rss.py:5-25) on the reference's shipped
scenarios and on synthetic traffic around the ego: per step and entity the record RSSDistances appended to its history
list and the safe (lateral, longitudinal) distances, and the two metric flags.  The callback and the metric are the
reference's own code; `intersects` of the hazard box with the safe buffer and its lines comes from the stand-in (exact
rational on the ego-frame coordinates the reference computed).  Only data is stored.
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
from scenario_gym.metrics.rss import RSS, RSSDistances  # noqa: E402
from scenario_gym.scenario import Scenario  # noqa: E402
from scenario_gym.xosc_interface import import_scenario  # noqa: E402

SCEN_DIR = "/root/reference/tests/input_files/Scenarios"
CODE = {"safe": 0, "lateral": 1, "longitudinal": 2, "both": 3, "unsafe_lateral": 4, "unsafe_longitudinal": 5}


def run(sc, dt):
    cb = RSSDistances()
    gym = ScenarioGym(timestep=dt, state_callbacks=[cb], metrics=[RSS()])
    gym.set_scenario(sc)
    ents = sc.entities
    codes, safes, ts = [], [], []

    def snap():
        row, srow = [], []
        for e in ents:
            if e is sc.ego or e not in cb.intersect:
                row.append(-1)
                srow.append([np.nan, np.nan])
                continue
            n = lens[e]
            hist = cb.intersect[e]
            if len(hist) == n:  # nothing appended this call (absent entity, or t == 0)
                row.append(-1)
                srow.append([np.nan, np.nan])
                continue
            lens[e] = len(hist)
            last = hist[-1]
            row.append(6 if isinstance(last, list) else CODE[last])  # the "found" case appends the list to itself
            sd = cb.safe_distances.get(e, [np.nan, np.nan])
            srow.append([float(sd[0]), float(sd[1])])
        codes.append(row)
        safes.append(srow)
        ts.append(gym.state.t)

    lens = {e: 1 for e in ents[1:]}
    snap()
    while not gym.state.is_done:
        gym.step()
        snap()
    m = gym.get_metrics()
    return np.array(ts), np.array(codes, np.int32), np.array(safes), bool(m["RSS_safe_longitudinal"]), bool(m["RSS_safe_lateral"])


def straight(p0, p1, t1, h):
    return np.array([[0.0, p0[0], p0[1], 0, h, 0, 0], [t1, p1[0], p1[1], 0, h, 0, 0]])


def main():
    out, names = {}, []
    for f in sorted(os.listdir(SCEN_DIR)):
        if not f.endswith(".xosc"):
            continue
        n = os.path.splitext(f)[0]
        sc = import_scenario(os.path.join(SCEN_DIR, f))
        if sc.entities[0] is not sc.ego or len(sc.entities) < 2:
            continue
        t, codes, safes, slong, slat = run(sc, 0.1)
        out.update(MG.flat(f"{n}/scenario", MG.export_scenario(sc)))
        out[f"{n}/t"], out[f"{n}/code"], out[f"{n}/safe"] = t, codes, safes
        out[f"{n}/safe_longitudinal"], out[f"{n}/safe_lateral"] = np.bool_(slong), np.bool_(slat)
        names.append(n)
        print(n, len(t), "long", slong, "lat", slat, np.bincount(codes[codes >= 0], minlength=7))
    rng = np.random.default_rng(12)
    for k in range(24):  # synthetic traffic: same-direction, oncoming and crossing vehicles close to the ego
        h0 = rng.uniform(-np.pi, np.pi)
        d, nrm = np.array([np.cos(h0), np.sin(h0)]), np.array([-np.sin(h0), np.cos(h0)])
        v_ego = rng.uniform(3, 15)
        ents = [MG.make_entity(straight(-0.5 * v_ego * 8 * d, 0.5 * v_ego * 8 * d, 8, h0), "ego", ctype="Vehicle")]
        for i in range(int(rng.integers(2, 6))):
            mode = rng.integers(0, 3)
            off = rng.uniform(-7, 7) * nrm + rng.uniform(-25, 25) * d
            if mode == 0:   # same direction, other speed, slight lateral drift
                v = rng.uniform(0, 18)
                dh = rng.normal(0, 0.04)
                e = np.array([np.cos(h0 + dh), np.sin(h0 + dh)])
                tr = straight(off - 4 * v * e, off + 4 * v * e, 8, h0 + dh)
            elif mode == 1:  # oncoming
                v = rng.uniform(3, 15)
                tr = straight(off + 4 * v * d, off - 4 * v * d, 8, h0 + np.pi)
            else:            # crossing
                h1 = h0 + rng.choice([-1, 1]) * rng.uniform(0.6, 2.4)
                e = np.array([np.cos(h1), np.sin(h1)])
                v = rng.uniform(2, 12)
                tr = straight(off - 4 * v * e, off + 4 * v * e, 8, h1)
            ents.append(MG.make_entity(tr, f"entity_{i}", ctype="Vehicle"))
        sc = Scenario(ents, name=f"synth{k}")
        n = f"synth{k}"
        t, codes, safes, slong, slat = run(sc, 0.1)
        out.update(MG.flat(f"{n}/scenario", MG.export_scenario(sc)))
        out[f"{n}/t"], out[f"{n}/code"], out[f"{n}/safe"] = t, codes, safes
        out[f"{n}/safe_longitudinal"], out[f"{n}/safe_lateral"] = np.bool_(slong), np.bool_(slat)
        names.append(n)
        print(n, len(t), "long", slong, "lat", slat, np.bincount(codes[codes >= 0], minlength=7))
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(os.environ.get("SG_GOLDEN_OUT", HERE), "rss.npz"), **out)


if __name__ == "__main__":
    main()
