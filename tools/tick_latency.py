import sys, time
sys.path.insert(0, '.')
import numpy as np
import scenario_gym_amd as sga
import scenario_gym_amd._lib as L
from scenario_gym_amd import synthetic
R, E = 4096, 64
packed = synthetic.make_batch(R, E, n_steps=2000, ego_kind=L.KIND_AGENT_VEHICLE)
eng = sga.RolloutEngine(R, E)
eng.upload(packed)
acts = synthetic.make_actions(1200, R)
for k in range(100):
    eng.step(1, acts[k:k+1])
t0 = time.perf_counter()
for k in range(100, 1100):
    eng.step(1, acts[k:k+1])
dt = (time.perf_counter() - t0) / 1000
print(f"tick (sg_step(1) with host actions, {R}x{E}): {dt*1e6:.1f} us -> {R*E/dt/1e9:.2f} G entity-steps/s")
t0 = time.perf_counter()
eng.step(1000, acts[100:1100])
dt = (time.perf_counter() - t0)
print(f"sg_step(1000) with host actions: {dt*1e3:.2f} ms -> {R*E*1000/dt/1e9:.2f} G entity-steps/s")
eng.close()

# FutureCollisionDetector look-ahead for the ego of every scenario
eng = sga.RolloutEngine(R, E)
eng.upload(packed)
eng.step(300)
eng.future_collision()
t0 = time.perf_counter()
for _ in range(50):
    f = eng.future_collision(5.0, 10)
dt = (time.perf_counter() - t0) / 50
print(f"sg_future_collision (horizon 5 s, 10 samples, {R}x{E}): {dt*1e6:.0f} us per call = {R*(E-1)*10/dt/1e9:.2f} G box pairs/s, "
      f"{int(f.sum())} of {R} scenarios flagged")
eng.close()

# RasterizedMapSensor "entity" layer around every ego
eng = sga.RolloutEngine(R, E)
eng.upload(packed)
eng.step(300)
for (w, n) in ((20.0, 20), (64.0, 128)):
    eng.raster_entities(w, w, n, n)
    t0 = time.perf_counter()
    for _ in range(10):
        m = eng.raster_entities(w, w, n, n)
    dt = (time.perf_counter() - t0) / 10
    print(f"sg_raster_entities ({n}x{n} over {w:.0f} m, {R}x{E}): {dt*1e3:.2f} ms per call incl. the {m.size/1e6:.1f} MB copy to the host "
          f"= {R*n*n/dt/1e9:.2f} G cells/s, {100*m.mean():.1f} % occupied")
eng.close()
