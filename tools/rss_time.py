"""Cost of running the RSS callback inside a rollout (sg_set_rss): python tools/rss_time.py [R] [E] [steps]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import scenario_gym_amd as sga
import scenario_gym_amd._lib as L
from scenario_gym_amd import synthetic

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
E = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=L.KIND_AGENT_PID)
for rss in (False, True):
    eng = sga.RolloutEngine(R, E)
    eng.set_rss(rss)
    eng.upload(packed)
    eng.rollout(steps)
    t0 = time.perf_counter()
    eng.rollout(steps)
    dt = time.perf_counter() - t0
    extra = ""
    if rss:
        sl, sa, codes, _ = eng.rss()
        extra = f"; safe_longitudinal in {int(sl.sum())} / {R} scenarios, safe_lateral in {int(sa.sum())}"
    print(f"rollout {R} x {E} x {steps} steps, RSS callback {'on' if rss else 'off'}: {dt*1e3:.1f} ms = {R*E*steps/dt/1e9:.2f} G entity-steps/s{extra}")
    eng.close()
