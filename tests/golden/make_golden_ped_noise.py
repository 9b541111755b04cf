#!/usr/bin/env python3
"""Golden vectors for the random fluctuations of the social force model (pedestrian/social_force.py:106-114): closed
loops of the REAL reference with non-zero std_lon / std_lat, the global numpy RNG seeded right before each rollout.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden_ped_noise.py      (build container only)

np.random.normal(loc, scale) draws from the legacy global stream: loc + scale * z, z = the next variate of
np.random.RandomState(seed).standard_normal -- one for the speed, one for the heading, per pedestrian that is still walking,
in agent order.  The file records the seed and the std so that the consumer can rebuild the same stream.  Only data is
written (scenario numbers and the reference's outputs).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

import make_golden as G  # noqa: E402  (sets up the import stand-ins and imports the reference)
from scenario_gym import ScenarioGym  # noqa: E402
from scenario_gym.agent import _create_agent  # noqa: E402
from scenario_gym.catalog_entry import BoundingBox, CatalogEntry  # noqa: E402
from scenario_gym.entity import Pedestrian  # noqa: E402
from scenario_gym.pedestrian.agent import PedestrianAgent  # noqa: E402
from scenario_gym.pedestrian.social_force import SocialForce, SocialForceParameters  # noqa: E402
from scenario_gym.road_network import RoadNetwork  # noqa: E402
from scenario_gym.scenario import Scenario  # noqa: E402
from scenario_gym.trajectory import Trajectory  # noqa: E402


def scene(seed, n_ped, side, n_wp, with_car):
    r = np.random.default_rng(seed)
    ents = []
    if with_car:
        ents.append(G.make_entity(np.array([[0.0, -side, 0.3, 0, 0.0, 0, 0], [9.0, side, 0.5, 0, 0.0, 0, 0]]), "ego", ctype="Vehicle"))
    routes, vds = {}, {}
    for i in range(n_ped):
        start = r.uniform(-side, side, 2)
        ce = CatalogEntry(None, "p", "p", "Pedestrian", BoundingBox(0.69, 0.7, 0.0, 0.0), {}, [])
        e = Pedestrian(ce, Trajectory(np.array([[0.0, *start, 0, r.uniform(-3, 3), 0, 0],
                                                [9.0, *(start + r.normal(0, 0.5, 2)), 0, 0.0, 0, 0]])), ref=f"ped_{i}")
        wps = [start + r.normal(0, 0.05, 2)]
        for _ in range(n_wp - 1):
            wps.append(-wps[-1] * r.uniform(0.3, 1.0) + r.normal(0, 1.0, 2))
        routes[e.ref] = np.array(wps)
        vds[e.ref] = r.uniform(0.5, 1.5) * 1.3
        ents.append(e)
    return Scenario(ents, name="crowd", road_network=RoadNetwork(roads=[], intersections=[])), routes, vds


def main():
    out = {}
    # (seed of the scene, pedestrians, half side, waypoints, sensor radius, car?, std_lon, std_lat, numpy seed)
    cases = [(11, 10, 2.5, 2, 3.0, True, 0.15, 0.08, 1234), (12, 24, 3.0, 3, 2.0, False, 0.3, 0.2, 99),
             (13, 70, 5.0, 2, 3.0, False, 0.000002, 0.0000001, 7)]  # the last one: the reference's default std
    for si, (seed, n_ped, side, n_wp, thr, car, std_lon, std_lat, np_seed) in enumerate(cases):
        sc, routes, vds = scene(seed, n_ped, side, n_wp, car)
        out.update(G.flat(f"loop{si}/scenario", G.export_scenario(sc)))
        refs = [e.ref for e in sc.entities]
        R = np.full((len(refs), n_wp, 2), np.nan)
        for k, ref in enumerate(refs):
            if ref in routes:
                R[k] = routes[ref]
        out[f"loop{si}/routes"] = R
        out[f"loop{si}/vdes"] = np.array([vds.get(ref, np.nan) for ref in refs])
        out[f"loop{si}/distance_threshold"] = np.float64(thr)
        out[f"loop{si}/noise"] = np.array([std_lon, std_lat, np_seed])

        def create_agent(s, e, routes=routes, vds=vds, thr=thr, std_lon=std_lon, std_lat=std_lat):
            if e.ref == "ego":
                return _create_agent(s, e)
            return PedestrianAgent(e, routes[e.ref], vds[e.ref], SocialForce(SocialForceParameters(std_lon=std_lon, std_lat=std_lat)),
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

        gym = ScenarioGym(timestep=1.0 / 30.0, metrics=G.std_metrics())
        gym.set_scenario(sc, create_agent=create_agent)
        np.random.seed(np_seed)  # the reference draws from the global RNG
        out.update(G.flat(f"loop{si}/dt30", G.record_rollout(gym, extra=extra)))
        # how many variates the rollout consumed: the stream position afterwards
        probe = np.random.RandomState(np_seed)
        nxt = np.random.standard_normal()
        stream = probe.standard_normal(400000)
        used = int(np.argmax(stream == nxt))
        assert stream[used] == nxt
        out[f"loop{si}/variates_used"] = np.int64(used)
    path = os.path.join(os.environ.get("SG_GOLDEN_OUT", HERE), "ped_noise.npz")
    np.savez_compressed(path, **out)
    print(f"ped_noise: {len(out)} arrays, {os.path.getsize(path) / 1e6:.2f} MB; variates used:",
          [int(out[f'loop{i}/variates_used']) for i in range(len(cases))])


if __name__ == "__main__":
    main()
