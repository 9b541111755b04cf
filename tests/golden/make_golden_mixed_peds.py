#!/usr/bin/env python3
"""Golden vectors for per-agent pedestrian behaviour models: closed loops of the REAL reference in which every
PedestrianAgent holds its OWN behaviour object (pedestrian/agent.py:18-41) -- SocialForce pedestrians of two different
parameter sets and RandomWalk pedestrians in one scenario -- with the global numpy RNG seeded right before each rollout.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden_mixed_peds.py      (build container only)

Every behaviour draws its two variates (speed, heading) from the one global generator, per pedestrian that is still walking,
in agent order, whatever its model (social_force.py:106-108, random_walk.py:37-43).  The file records, per pedestrian, the
model it follows and, per model, behaviour / parameters / std so that the consumer can rebuild the same loop.  Only data is
written (scenario numbers and the reference's outputs).
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
from scenario_gym.pedestrian.social_force import SocialForce, SocialForceParameters  # noqa: E402

# model rows: behaviour (0 SocialForce / 1 RandomWalk), relaxation_time, ped_repulse_V, ped_repulse_sigma, ped_attract_C,
# sight_weight, sight_weight_use, sight_angle, max_speed_factor, bias_lon, bias_lat, std_lon, std_lat
MODEL_COLS = ["behaviour", "relaxation_time", "ped_repulse_V", "ped_repulse_sigma", "ped_attract_C", "sight_weight",
              "sight_weight_use", "sight_angle", "max_speed_factor", "bias_lon", "bias_lat", "std_lon", "std_lat"]


def behaviour_of(row):
    m = dict(zip(MODEL_COLS, row))
    if m["behaviour"] == 1:
        return RandomWalk(RandomWalkParameters(bias_lon=m["bias_lon"], bias_lat=m["bias_lat"], std_lon=m["std_lon"], std_lat=m["std_lat"],
                                               max_speed_factor=m["max_speed_factor"]))
    return SocialForce(SocialForceParameters(
        relaxation_time=m["relaxation_time"], ped_repulse_V=m["ped_repulse_V"], ped_repulse_sigma=m["ped_repulse_sigma"],
        ped_attract_C=m["ped_attract_C"], sight_weight=m["sight_weight"], sight_weight_use=bool(m["sight_weight_use"]),
        sight_angle=m["sight_angle"], max_speed_factor=m["max_speed_factor"], bias_lon=m["bias_lon"], bias_lat=m["bias_lat"],
        std_lon=m["std_lon"], std_lat=m["std_lat"]))


def main():
    out = {"model_cols": np.array(MODEL_COLS)}
    sfA = [0, 1.5, 1.0, 1.0, 0.0, 0.5, 1, 200, 1.3, 0.0, 0.0, 0.1, 0.1]     # the reference's defaults
    sfB = [0, 0.8, 2.5, 0.6, 0.0, 0.3, 1, 160, 1.1, 0.05, -0.02, 0.02, 0.3]  # a second SocialForce parameter set
    sfC = [0, 1.2, 1.0, 1.4, 0.02, 0.5, 0, 200, 1.3, 0.0, 0.0, 0.0, 0.0]     # attraction on, sight weights off, std 0
    rw = [1, 0, 0, 0, 0, 0, 0, 0, 1.3, 0.1, 0.05, 0.3, 0.2]                  # RandomWalk
    rw0 = [1, 0, 0, 0, 0, 0, 0, 0, 1.3, 0.0, 0.0, 0.000002, 0.0000001]       # RandomWalk, the reference's defaults
    # (seed of the scene, pedestrians, half side, waypoints, car?, models, pattern of model indices over the pedestrians, numpy seed)
    cases = [(31, 12, 2.5, 3, True, [sfA, rw], [0, 1], 99),
             (32, 24, 3.5, 3, False, [sfA, sfB, rw], [0, 1, 2, 1, 0], 7),
             (33, 16, 2.2, 2, False, [sfB, sfC, rw0, rw], [0, 1, 2, 3], 123),
             (34, 10, 3.0, 3, False, [sfA, sfB], [0, 1], 5)]                  # two SocialForce sets, no RandomWalk
    for si, (seed, n_ped, side, n_wp, car, models, pattern, np_seed) in enumerate(cases):
        sc, routes, vds = N.scene(seed, n_ped, side, n_wp, car)
        out.update(G.flat(f"loop{si}/scenario", G.export_scenario(sc)))
        refs = [e.ref for e in sc.entities]
        R = np.full((len(refs), n_wp, 2), np.nan)
        model_of = np.full(len(refs), -1, np.int32)
        k_ped = 0
        for k, ref in enumerate(refs):
            if ref in routes:
                R[k] = routes[ref]
                model_of[k] = pattern[k_ped % len(pattern)]
                k_ped += 1
        out[f"loop{si}/routes"] = R
        out[f"loop{si}/vdes"] = np.array([vds.get(ref, np.nan) for ref in refs])
        out[f"loop{si}/models"] = np.array(models, np.float64)
        out[f"loop{si}/model_of"] = model_of
        out[f"loop{si}/np_seed"] = np.int64(np_seed)
        behaviours = [behaviour_of(row) for row in models]  # (agents of one model share one behaviour object: it is stateless)
        midx = dict(zip(refs, model_of))

        def create_agent(s, e, routes=routes, vds=vds, behaviours=behaviours, midx=midx):
            if e.ref == "ego":
                return _create_agent(s, e)
            return PedestrianAgent(e, routes[e.ref], vds[e.ref], behaviours[midx[e.ref]])

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
    path = os.path.join(os.environ.get("SG_GOLDEN_OUT", HERE), "mixed_peds.npz")
    np.savez_compressed(path, **out)
    print(f"mixed_peds: {len(out)} arrays, {os.path.getsize(path) / 1e6:.2f} MB; variates used:",
          [int(out[f'loop{i}/variates_used']) for i in range(len(cases))])


if __name__ == "__main__":
    main()
