#!/usr/bin/env python3
"""Golden vectors for the observation kernels from the REAL reference: tests/golden/sensors.npz.

Build container only (needs /root/reference and the import stand-ins of tests/golden/_refstubs, see its README;
collision predicates come from the stand-in's exact-rational SAT on the reference's own fp64 corners):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden_sensors.py

Records, for shipped XOSC scenarios rolled out by the reference at dt = 1/30 and 0.1, FutureCollisionDetector(ego)
(sensor/common.py:59-106; horizons 5.0 and 1.0) evaluated on the state after reset and after every step, and the "entity"
layer of RasterizedMapSensor(ego) (sensor/map.py:136-192) in two grid configurations on every 4th state of the dt = 0.1
rollouts.  Only data.
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
from scenario_gym.sensor.common import FutureCollisionDetector  # noqa: E402
from scenario_gym.sensor.map import RasterizedMapSensor  # noqa: E402
from scenario_gym.xosc_interface import import_scenario  # noqa: E402

assert scenario_gym.__version__ == "0.3.1"
SCEN_DIR = "/root/reference/tests/input_files/Scenarios"
NAMES = ["a5e43fe4", "3fee6507", "41dac6fa", "5c5188e0", "a98d5c7d"]  # the scenarios exported in scenarios.npz


def main():
    out = {"names": np.array(NAMES), "horizons": np.array([5.0, 1.0]),
           # (width, height, n per side): freq = 1 over 30 m x 30 m, and a fine 24 x 24 grid over 12 m x 12 m
           "raster_cfg": np.array([[30.0, 30.0, 30.0], [12.0, 12.0, 24.0]])}
    for n in NAMES:
        path = [os.path.join(SCEN_DIR, f) for f in sorted(os.listdir(SCEN_DIR)) if f.startswith(n)][0]
        for dtn, dt in (("dt30", 1 / 30), ("dt10", 0.1)):
            gym = ScenarioGym(timestep=dt)
            gym.set_scenario(import_scenario(path))
            ego = gym.state.scenario.entities[0]
            sensors = [FutureCollisionDetector(ego, horizon=h) for h in out["horizons"]]
            rasters = [RasterizedMapSensor(ego, layers=["entity"], width=w, height=h, freq=None, n=int(n))
                       for w, h, n in out["raster_cfg"]]
            ts, flags = [gym.state.t], [[s.reset(gym.state).future_collision for s in sensors]]
            maps = [[np.asarray(r.reset(gym.state).map)[:, :, 0]] for r in rasters]
            map_steps = [0]
            while not gym.state.is_done:
                gym.step()
                ts.append(gym.state.t)
                flags.append([s.step(gym.state).future_collision for s in sensors])
                if dtn == "dt10" and (len(ts) - 1) % 4 == 0:
                    map_steps.append(len(ts) - 1)
                    for m, r in zip(maps, rasters):
                        m.append(np.asarray(r.step(gym.state).map)[:, :, 0])
            if dtn == "dt10":
                out[f"{n}/{dtn}/map_steps"] = np.array(map_steps)
                for k, m in enumerate(maps):
                    out[f"{n}/{dtn}/map{k}"] = np.array(m, np.uint8)  # [frames][n][n]
                print(n, "maps", [int(np.array(m).sum()) for m in maps])
            out[f"{n}/{dtn}/t"] = np.array(ts)
            out[f"{n}/{dtn}/future"] = np.array(flags, np.uint8)  # [steps + 1][horizon]
            print(n, dtn, len(ts), np.array(flags).sum(0))
    np.savez_compressed(os.path.join(os.environ.get("SG_GOLDEN_OUT", HERE), "sensors.npz"), **out)


if __name__ == "__main__":
    main()
