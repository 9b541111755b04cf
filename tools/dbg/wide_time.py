"""Throughput of the multi-kernel step (scenarios of more than 512 entities)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scenario_gym_amd as sga
import scenario_gym_amd._lib as L
from scenario_gym_amd import synthetic
for R, E, steps, what in ((64, 1024, 300, "vehicles"), (256, 1024, 300, "vehicles"), (16, 4096, 100, "vehicles"), (64, 1024, 200, "crowd")):
    packed = synthetic.make_crowd(R, E, n_steps=steps, side=40.0 * (E / 256) ** 0.5) if what == "crowd" else synthetic.make_batch(R, E, n_steps=steps, ego_kind=L.KIND_AGENT_PID, extent=100.0 * (E / 64) ** 0.5)  # the density of the c3 batch
    eng = sga.RolloutEngine(R, E)
    eng.upload(packed)
    eng.rollout(steps)
    t = time.perf_counter(); eng.rollout(steps); eng.synchronize(); dt = time.perf_counter() - t
    n = int(eng.state()["n_steps"].sum())
    print(f"{what} R={R} E={E}: {dt * 1e3:.1f} ms for {steps} steps = {dt / steps * 1e6:.0f} us per step, {n * E / dt / 1e9:.3f} G entity-steps/s")
    eng.close()
