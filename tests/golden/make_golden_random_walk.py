#!/usr/bin/env python3
"""Golden vectors for the RandomWalk behaviour (pedestrian/random_walk.py:22-44): closed loops of the REAL reference --
PedestrianAgent(..., behaviour=RandomWalk(params)) -- with the global numpy RNG seeded right before each rollout.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden_random_walk.py      (build container only)

RandomWalk._step draws np.random.normal(speed_desired + bias_lon, std_lon) and np.random.normal(angle + bias_lat, std_lat),
in that order, per pedestrian that is still walking, in agent order: loc + scale * z with z the next variates of
np.random.RandomState(seed).standard_normal.  The file records seed, std and bias so that the consumer can rebuild the same
stream.  Only data is written (scenario numbers and the reference's outputs).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

import make_golden as G  # noqa: E402  (sets up the import stand-ins and imports the reference)
import make_golden_ped_noise as N  # noqa: E402  (the scenes)
from scenario_gym import ScenarioGym  # noqa: E402
from scenario_gym.agent import _create_agent  # noqa: E402
from scenario_gym.pedestrian.agent import PedestrianAgent  # noqa: E402
from scenario_gym.pedestrian.random_walk import RandomWalk, RandomWalkParameters  # noqa: E402


def main():
    out = {}
    # (seed of the scene, pedestrians, half side, waypoints, car?, std_lon, std_lat, bias_lon, bias_lat, max_speed, numpy seed)
    cases = [(21, 9, 2.5, 3, True, 0.2, 0.3, 0.0, 0.0, 5.0, 4321),
             (22, 30, 4.0, 4, False, 0.5, 0.1, 0.25, -0.05, 1.6, 17),            # a bias, and a max_speed that clips
             (23, 12, 3.0, 2, False, 0.000002, 0.0000001, 0.0, 0.0, 5.0, 5),     # the reference's default parameters
             (24, 8, 3.0, 3, False, 0.0, 0.0, 0.0, 0.0, 5.0, 1)]                 # std 0: normal(loc, 0) == loc
    for si, (seed, n_ped, side, n_wp, car, std_lon, std_lat, bias_lon, bias_lat, max_speed, np_seed) in enumerate(cases):
        sc, routes, vds = N.scene(seed, n_ped, side, n_wp, car)
        out.update(G.flat(f"loop{si}/scenario", G.export_scenario(sc)))
        refs = [e.ref for e in sc.entities]
        R = np.full((len(refs), n_wp, 2), np.nan)
        for k, ref in enumerate(refs):
            if ref in routes:
                R[k] = routes[ref]
        out[f"loop{si}/routes"] = R
        out[f"loop{si}/vdes"] = np.array([vds.get(ref, np.nan) for ref in refs])
        out[f"loop{si}/params"] = np.array([std_lon, std_lat, bias_lon, bias_lat, max_speed, np_seed])

        def create_agent(s, e, routes=routes, vds=vds, pr=(std_lon, std_lat, bias_lon, bias_lat), max_speed=max_speed):
            if e.ref == "ego":
                return _create_agent(s, e)
            params = RandomWalkParameters(std_lon=pr[0], std_lat=pr[1], bias_lon=pr[2], bias_lat=pr[3])
            return PedestrianAgent(e, routes[e.ref], vds[e.ref], RandomWalk(params), max_speed=max_speed)

        def extra(g):
            rows = []
            for e in g.state.scenario.entities:
                a = g.state.agents.get(e)
                if isinstance(a, PedestrianAgent):
                    rows.append([a.controller.speed, float(a.goal_idx), a.force[0], a.force[1]])
                else:
                    rows.append([np.nan] * 4)
            return rows

        gym = ScenarioGym(timestep=1.0 / 30.0, metrics=G.std_metrics())
        gym.set_scenario(sc, create_agent=create_agent)
        np.random.seed(np_seed)  # the reference draws from the global RNG
        out.update(G.flat(f"loop{si}/dt30", G.record_rollout(gym, extra=extra)))
        probe = np.random.RandomState(np_seed)  # how many variates the rollout consumed
        nxt = np.random.standard_normal()
        stream = probe.standard_normal(400000)
        used = int(np.argmax(stream == nxt))
        assert stream[used] == nxt
        out[f"loop{si}/variates_used"] = np.int64(used)
    path = os.path.join(os.environ.get("SG_GOLDEN_OUT", HERE), "random_walk.npz")
    np.savez_compressed(path, **out)
    print(f"random_walk: {len(out)} arrays, {os.path.getsize(path) / 1e6:.2f} MB; variates used:",
          [int(out[f'loop{i}/variates_used']) for i in range(len(cases))])


if __name__ == "__main__":
    main()
