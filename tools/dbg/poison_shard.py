"""The `strong` shard of rank 1 of `bench.py --gpus 2 --scenarios 1024 --sim-steps 2000`, alone in one process, with poisoned allocations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scenario_gym_amd as sga
import scenario_gym_amd._lib as L
from scenario_gym_amd import synthetic
from oracle import check

T, dt = 2000, 1 / 30
for first, R in ((512, 512), (0, 512), (1024, 1024), (512, 512)):
    packed = synthetic.make_batch(R, 64, n_steps=T, timestep=dt, ego_kind=L.KIND_AGENT_PID, seed=int(os.environ.get("SEED", "0")) or 2024, first_scenario=first)
    eng = sga.RolloutEngine(R, 64, timestep=dt, terminal_conditions=["max_length"], event_capacity=64)
    eng.set_slicing(False)
    eng.upload(packed)
    for rep in range(3):
        eng.rollout_async(T, do_reset=True)
        eng.synchronize()
        v = check.verify_engine(eng, packed, dt, T, K=4, event_cap=64)
        print(f"first {first} R {R} pass {rep}: equal {v['equal']} {v['mismatches']} schedule {eng.schedule_info()['schedule']}", flush=True)
    eng.close()
