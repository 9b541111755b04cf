#!/usr/bin/env python3
"""Time the REAL reference (driskai/scenario_gym, imported from /root/reference with the import stand-ins of
tests/golden/_refstubs) on scenarios of the bench's synthetic family -- SURVEY.md 8(d)(i).  Build container only.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tools/time_reference.py [--entities 64] [--steps 1000] [--scenarios 2]

One process per core (the reference is single-threaded), each rolling `--scenarios` scenarios of E entities for `--steps`
steps with the default agents (ego ReplayTrajectoryAgent, others batch replay) and the three ego metrics; collisions are
left out of the timed runs because the stand-in's exact-rational SAT is far slower than GEOS and says nothing about the
reference.  Prints entity-steps/s per core and for the box, and writes profiles/reference_cpu.json.
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
    seed_chunk, n_scen, E, steps = args
    sys.path[:0] = [os.path.join(ROOT, "tests", "golden", "_refstubs"), "/root/reference", ROOT]
    import numpy as np
    from scenario_gym import ScenarioGym
    from scenario_gym.catalog_entry import BoundingBox, CatalogEntry
    from scenario_gym.entity import Entity
    from scenario_gym.metrics import EgoAvgSpeed, EgoDistanceTravelled, EgoMaxSpeed
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
    gym = ScenarioGym(timestep=dt, metrics=[EgoAvgSpeed(), EgoMaxSpeed(), EgoDistanceTravelled()])
    t0 = time.perf_counter()
    n = 0
    for sc in scenarios:
        gym.set_scenario(sc)
        gym.rollout()
        n += round((gym.state.t - max(0.0, sc.ego.trajectory.min_t)) / dt) * E
    return n, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--entities", type=int, default=64)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--scenarios", type=int, default=2, help="per process")
    ap.add_argument("--procs", type=int, default=os.cpu_count() or 1)
    a = ap.parse_args()
    with mp.get_context("spawn").Pool(a.procs) as pool:
        t0 = time.perf_counter()
        res = pool.map(_worker, [(k, a.scenarios, a.entities, a.steps) for k in range(a.procs)])
        wall = time.perf_counter() - t0
    per_core = [n / t for n, t in res]
    import numpy, scipy

    out = dict(entities=a.entities, sim_steps=a.steps, scenarios_per_process=a.scenarios, processes=a.procs,
               entity_steps_per_s_per_core=sum(per_core) / len(per_core), entity_steps_per_s_box=sum(n for n, _ in res) / max(t for _, t in res),
               wall_s=wall, numpy=numpy.__version__, scipy=scipy.__version__,
               note="reference v0.3.1 imported from /root/reference with tests/golden/_refstubs; default agents + 3 ego metrics, "
                    "no collision metric; one single-threaded process per core of the build container")
    print(json.dumps(out, indent=1))
    json.dump(out, open(os.path.join(ROOT, "profiles", "reference_cpu.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
