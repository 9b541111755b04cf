#!/usr/bin/env python3
"""1024 x 256 crowd whose pedestrians follow three SocialForce parameter sets: rollout_kernel_crowd_models<4> against the general
pedestrian variant (SG_CROWD_MODELS=0), ms per 10,000-step rollout.  GPU box."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import numpy as np

    import scenario_gym_amd as sga
    from scenario_gym_amd import synthetic

    R, E, T = 1024, 256, 10000
    packed = synthetic.make_crowd(R, E, n_steps=T)
    models = [dict(), dict(relaxation_time=0.8, ped_repulse_V=2.5, ped_repulse_sigma=0.6, sight_weight=0.3), dict(relaxation_time=0.3, ped_repulse_sigma=0.35)]
    eng = sga.RolloutEngine(R, E, timestep=1 / 30, terminal_conditions=["max_length"], event_capacity=64)
    eng.set_ped_models(models, np.random.default_rng(1).integers(0, 3, R * E).astype(np.int32))
    eng.upload(packed)
    eng.rollout(T)
    t0 = time.perf_counter()
    eng.rollout(T)
    dt = time.perf_counter() - t0
    print(f"{eng.last_kernel()}: {dt * 1e3:.1f} ms, {R * E * T / dt / 1e9:.2f} G entity-steps/s")
else:
    for v in ("1", "0"):
        subprocess.run([sys.executable, __file__, "run"], env=dict(os.environ, SG_CROWD_MODELS=v))
