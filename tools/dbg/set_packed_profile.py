"""Where the time of BatchedScenarioGym.set_packed goes (the e2e line's `upload` stage): per-call times over a run of chunks."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import scenario_gym_amd as sga
from scenario_gym_amd import metrics as M, synthetic
import scenario_gym_amd._lib as L

packed = synthetic.make_batch(512, 40, n_steps=330, n_knots=111, ego_kind=L.KIND_AGENT_REPLAY)
packed.refs = [[f"e{k}" for k in range(40)] for _ in range(512)]
gym = sga.BatchedScenarioGym(timestep=1 / 30, state_callbacks=[M.RSSDistances()],
                             metrics=lambda: [M.EgoAvgSpeed(), M.EgoMaxSpeed(), M.EgoDistanceTravelled(), M.CollisionMetric(), M.RSS()], event_capacity=16)
gym.set_packed(packed); gym.rollout(); gym.get_metrics()
import gc
for mode in ("upload only", "upload + rollout + metrics", "gc off: upload only"):
    if mode.startswith("gc off"):
        gc.disable()
    ts = []
    for _ in range(24):
        t = time.perf_counter()
        gym.set_packed(packed)
        t1 = time.perf_counter()
        if mode != "upload only":
            gym.rollout(); gym.engine.synchronize(); gym.get_metrics()
        ts.append((t1 - t) * 1e3)
    print(mode, "set_packed ms:", " ".join(f"{x:.1f}" for x in ts))
