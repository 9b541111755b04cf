#!/usr/bin/env python3
"""Golden vectors for the boundary terms of the social force from the REAL reference:
tests/golden/ped_roads.npz.

Build container only (needs /root/reference and the import stand-ins of tests/golden/_refstubs, see its README):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden_ped_roads.py

The road network is the one `examples/crowds.py:149-205` builds (one road with two lanes, two pavements, one building),
the crowd walks along the pavement beside the building.  SocialForce._step and everything around it is the reference's
code; `nearest_points`, `contains` and `area` come from the stand-in (GEOS DistanceOp / RayCrossingCounter restated,
see the README), so the boundary forces are pinned against that restatement, not against GEOS.  Only data is stored.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path[:0] = [HERE, os.path.join(HERE, "_refstubs"), "/root/reference"]

import numpy as np  # noqa: E402

import make_golden as MG  # noqa: E402  (helpers: export_scenario, flat, record_rollout, std_metrics)
from make_golden_roads import export_network  # noqa: E402
from scenario_gym import ScenarioGym  # noqa: E402
from scenario_gym.catalog_entry import BoundingBox, CatalogEntry  # noqa: E402
from scenario_gym.entity import Pedestrian  # noqa: E402
from scenario_gym.pedestrian.agent import PedestrianAgent  # noqa: E402
from scenario_gym.pedestrian.social_force import SocialForce, SocialForceParameters  # noqa: E402
from scenario_gym.road_network import Building, Lane, Pavement, Road, RoadNetwork  # noqa: E402
from scenario_gym.scenario import Scenario  # noqa: E402
from scenario_gym.scenario_gym import _create_agent  # noqa: E402
from scenario_gym.trajectory import Trajectory  # noqa: E402
from shapely.geometry import LineString, Polygon  # noqa: E402  (stand-in)


def boundary(w0, w1, l0, l1):
    return Polygon(np.array([[w0, l0], [w0, l1], [w1, l1], [w1, l0]]))


def center(w, l0, l1):
    return LineString(np.array([[w, l0], [w, l1]]))


def road_network(w=4.2, l=25):
    """examples/crowds.py:149-205."""
    lanes = [Lane("lane_0", boundary(-w, 0, -l, l), center(-w / 2, -l, l), [], [], "driving"),
             Lane("lane_1", boundary(0, w, -l, l), center(w / 2, -l, l), [], [], "driving")]
    road = Road("road_0", boundary(-w, w, -l, l), center(0, -l, l), lanes=lanes)
    pavements = [Pavement("pave_0", boundary(-w - 2, -w, -l, l), center(-w - 1, -l, l)),
                 Pavement("pave_1", boundary(w, w + 1, -l, l), center(w + 1 / 2, -l, l))]
    building = Building("building_0", Polygon(np.array([[-10.0, 0.0], [-10.0, 10.0], [-5.2, 10.0], [-5.2, 0.0]])))
    return RoadNetwork(roads=[road], intersections=[], pavements=pavements, buildings=[building])


def main():
    out = {}
    rn = road_network()
    for k, v in export_network(rn).items():
        out[f"net/{k}"] = v
    for si, (seed, n_ped, thr) in enumerate([(5, 14, 3.0), (6, 22, 1.0)]):
        r = np.random.default_rng(seed)
        ents = [MG.make_entity(np.array([[0.0, 2.0, -20.0, 0, 1.57, 0, 0], [12.0, 2.0, 20.0, 0, 1.57, 0, 0]]), "ego", ctype="Vehicle")]
        routes, vds = {}, {}
        for i in range(n_ped):
            # starts on pave_0 (x in [-6.2, -4.2]) and in the lane beside it, some hugging the building wall x = -5.2
            start = np.array([r.uniform(-5.15, -3.0), r.uniform(-6.0, 16.0)])
            if i % 5 == 0:
                start = np.array([r.uniform(-5.19, -5.0), r.uniform(1.0, 9.0)])
            if i % 7 == 3:
                start = np.array([r.uniform(-9.0, -6.0), r.uniform(2.0, 8.0)])  # inside the building: zero force, sign -1
            goal = np.array([r.uniform(-6.0, -4.4), start[1] + r.choice([-1, 1]) * r.uniform(6.0, 14.0)])
            ce = CatalogEntry(None, "p", "p", "Pedestrian", BoundingBox(0.69, 0.7, 0.0, 0.0), {}, [])
            e = Pedestrian(ce, Trajectory(np.array([[0.0, *start, 0, r.uniform(-3, 3), 0, 0], [12.0, *goal, 0, 0.0, 0, 0]])),
                           ref=f"ped_{i}")
            routes[e.ref] = np.array([start + r.normal(0, 0.05, 2), goal])
            vds[e.ref] = r.uniform(0.5, 1.5) * 1.3
            ents.append(e)
        sc = Scenario(ents, name="crowd", road_network=rn)
        out.update(MG.flat(f"loop{si}/scenario", MG.export_scenario(sc)))
        refs = [e.ref for e in sc.entities]
        R = np.full((len(refs), 2, 2), np.nan)
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
                    rows.append([a.controller.speed, float(a.goal_idx), a.force[0], a.force[1]]
                                if isinstance(a, PedestrianAgent) else [np.nan] * 4)
                return rows

            gym = ScenarioGym(timestep=dt, metrics=MG.std_metrics())
            gym.set_scenario(sc, create_agent=create_agent)
            rec = MG.record_rollout(gym, extra=extra)
            out.update(MG.flat(f"loop{si}/{dt_name}", rec))
            print(si, dt_name, len(rec["t"]), "max |force|", float(np.nanmax(np.abs(np.array(rec["extra"])[:, :, 2:]))))
    np.savez_compressed(os.path.join(os.environ.get("SG_GOLDEN_OUT", HERE), "ped_roads.npz"), **out)


if __name__ == "__main__":
    main()
