#!/usr/bin/env python3
"""Time the REAL reference (driskai/scenario_gym, imported from /root/reference with the import stand-ins of
tests/golden/_refstubs) on scenarios of the bench's synthetic family -- SURVEY.md 8(d)(i).  Build container only.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tools/time_reference.py [--entities 64] [--steps 1000] [--scenarios 2]

One process per core (the reference is single-threaded), each rolling `--scenarios` scenarios of E entities for `--steps`
steps.  `--ego pid` (the default; BASELINE config 3 / SURVEY 8d(i): 8 processes x 2 replicas x 64 entities x 10,000 steps)
gives the ego a PIDAgent with the default gains (agent.py:131-148, controller.py:157-161), the others stay batch replay;
`--ego replay` is the default-agent run (ego ReplayTrajectoryAgent).  The three ego metrics are always on.  `--collisions`
adds CollisionMetric: State.collisions() then goes through the shapely STAND-IN (exact-rational SAT, far slower than GEOS),
so that figure says nothing about the reference and is recorded separately, as SURVEY 8d(i) asks.
Prints entity-steps/s per core and for the box, and writes profiles/reference_cpu[_<tag>].json.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")


def _worker(args):
    seed_chunk, n_scen, E, steps, ego, collisions = args
    sys.path[:0] = [os.path.join(ROOT, "tests", "golden", "_refstubs"), "/root/reference", ROOT]
    import numpy as np
    from scenario_gym import ScenarioGym
    from scenario_gym.catalog_entry import BoundingBox, CatalogEntry
    from scenario_gym.entity import Entity
    from scenario_gym.agent import PIDAgent, _create_agent
    from scenario_gym.metrics import CollisionMetric, EgoAvgSpeed, EgoDistanceTravelled, EgoMaxSpeed
    from scenario_gym.scenario import Scenario
    from scenario_gym.trajectory import Trajectory

    from scenario_gym_amd import synthetic
    from scenario_gym_amd.packing import unpack_scenario

    dt = 1.0 / 30.0
    packed = synthetic.make_batch(synthetic.CHUNK, E, n_steps=steps, timestep=dt, first_scenario=seed_chunk * synthetic.CHUNK)
    scenarios = []
    for r in range(n_scen):
        s = unpack_scenario(packed, r)
        ents = []
        for e in range(E):
            bb = BoundingBox(*[float(x) for x in s["bbox"][e]])
            ce = CatalogEntry("synthetic", "car1", "car", "Vehicle", bb, {}, [])
            ents.append(Entity(ce, ref="ego" if e == 0 else f"vehicle_{e - 1}",
                               trajectory=Trajectory(s["knots"][s["knot_off"][e]:s["knot_off"][e + 1]])))
        scenarios.append(Scenario(ents, name=f"synthetic_{seed_chunk}_{r}"))
    metrics = [EgoAvgSpeed(), EgoMaxSpeed(), EgoDistanceTravelled()] + ([CollisionMetric()] if collisions else [])
    gym = ScenarioGym(timestep=dt, metrics=metrics)

    def create_agent(sc, e):  # the bench's c3: a PIDAgent (default gains) drives the ego along its own trajectory
        if ego == "pid" and e.ref == "ego":
            return PIDAgent(e)
        return _create_agent(sc, e)

    t0 = time.perf_counter()
    n = 0
    for sc in scenarios:
        gym.set_scenario(sc, create_agent=create_agent)
        gym.rollout()
        n += round((gym.state.t - max(0.0, sc.ego.trajectory.min_t)) / dt) * E
    return n, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--entities", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10000)
    ap.add_argument("--ego", choices=["pid", "replay"], default="pid")
    ap.add_argument("--collisions", action="store_true", help="add CollisionMetric (runs the shapely stand-in: not representative)")
    ap.add_argument("--tag", default="", help="suffix of the output file")
    ap.add_argument("--scenarios", type=int, default=2, help="per process")
    ap.add_argument("--procs", type=int, default=os.cpu_count() or 1)
    a = ap.parse_args()
    with mp.get_context("spawn").Pool(a.procs) as pool:
        t0 = time.perf_counter()
        res = pool.map(_worker, [(k, a.scenarios, a.entities, a.steps, a.ego, a.collisions) for k in range(a.procs)])
        wall = time.perf_counter() - t0
    per_core = [n / t for n, t in res]
    import numpy, scipy

    out = dict(config="BASELINE configs[2] (c3)" if (a.ego == "pid" and a.entities == 64 and a.steps == 10000) else "other",
               ego=a.ego, collision_metric=bool(a.collisions), entities=a.entities, sim_steps=a.steps, scenarios_per_process=a.scenarios, processes=a.procs,
               entity_steps_per_s_per_core=sum(per_core) / len(per_core), entity_steps_per_s_box=sum(n for n, _ in res) / max(t for _, t in res),
               wall_s=wall, numpy=numpy.__version__, scipy=scipy.__version__,
               entity_steps_total=sum(n for n, _ in res), per_process_s=[round(t, 2) for _, t in res],
               note="reference v0.3.1 imported from /root/reference with tests/golden/_refstubs (stand-ins for the absent lxml / "
                    "shapely); one single-threaded process per core of the 8-core build container; 3 ego metrics"
                    + (", CollisionMetric through the shapely stand-in (NOT GEOS: not representative)" if a.collisions
                       else ", no collision metric (GEOS is absent here)"))
    print(json.dumps(out, indent=1))
    json.dump(out, open(os.path.join(ROOT, "profiles", f"reference_cpu{'_' + a.tag if a.tag else ''}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
