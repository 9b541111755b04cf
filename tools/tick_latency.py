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

# RasterizedMapSensor with road layers (sg_raster_map) over a synthetic road network shared by all scenarios, and the
# RL tick of integrations/openaigym.py: one step with host actions + the default observation (entity + driveable_surface)
rng = np.random.default_rng(1)
rings = []
for q in range(60):  # 60 "roads": blobs of 40-120 vertices over the +-120 m the synthetic egos roam
    c = rng.uniform(-120, 120, 2)
    m = int(rng.integers(40, 120))
    ang = np.sort(rng.uniform(0, 2 * np.pi, m))
    rad = rng.uniform(15, 50) * rng.uniform(0.7, 1.0, m)
    rings.append(c + rad[:, None] * np.stack([np.cos(ang), np.sin(ang)], 1))
net = dict(ring_off=np.arange(len(rings) + 1), vert_off=np.concatenate([[0], np.cumsum([len(r) for r in rings])]),
           verts=np.concatenate(rings), layers=rng.choice([1 | 2, 1 | 4, 1 | 8, 16 | 32, 16 | 64], len(rings)))
eng = sga.RolloutEngine(R, E)
eng.upload(packed)
t0 = time.perf_counter()
eng.set_road_networks([net], np.zeros(R, np.int32))
print(f"sg_set_road_networks ({len(rings)} polygons, {len(net['verts'])} vertices): {(time.perf_counter() - t0)*1e3:.1f} ms")
eng.step(300)
for layers, n, w in (([0, 1], 20, 20.0), ([0, 1, 2, 4, 8, 16, 32, 64], 61, 30.0)):
    eng.raster_map(layers, w, w, n, n)
    t0 = time.perf_counter()
    for _ in range(10):
        m = eng.raster_map(layers, w, w, n, n)
    dt = (time.perf_counter() - t0) / 10
    print(f"sg_raster_map ({len(layers)} layers, {n}x{n} over {w:.0f} m, {R}x{E}): {dt*1e3:.2f} ms per call incl. the {m.size/1e6:.1f} MB "
          f"copy to the host = {R*n*n*len(layers)/dt/1e9:.2f} G cells/s, driveable {100*m[:, 1].mean():.1f} %")
for k in range(20):
    eng.step(1, acts[k:k+1]); eng.raster_map([0, 1])
t0 = time.perf_counter()
for k in range(20, 520):
    eng.step(1, acts[k:k+1])
    obs = eng.raster_map([0, 1])
dt = (time.perf_counter() - t0) / 500
print(f"RL tick (sg_step(1) with host actions + default 20x20x2 map observation, {R} envs): {dt*1e6:.0f} us = {R/dt/1e6:.1f} M env-steps/s")
eng.close()

# the same tick with policy and observation on the GPU: actions from a torch tensor (device pointer), map left in HBM
import torch
eng = sga.RolloutEngine(R, E)
eng.upload(packed)
eng.set_road_networks([net], np.zeros(R, np.int32))
eng.step(300)
act_t = torch.as_tensor(acts[:600], device="cuda:0")
for k in range(20):
    eng.step(1, act_t[k:k+1]); eng.raster_map_torch([0, 1])
t0 = time.perf_counter()
for k in range(20, 520):
    eng.step(1, act_t[k:k+1])
    obs = eng.raster_map_torch([0, 1])
dt = (time.perf_counter() - t0) / 500
print(f"RL tick, device-resident (actions from a torch tensor, 20x20x2 map as a torch view, {R} envs): {dt*1e6:.0f} us = "
      f"{R/dt/1e6:.1f} M env-steps/s; obs {tuple(obs.shape)} {obs.dtype}, driveable {100*obs[:, 1].float().mean().item():.1f} %")
eng.close()

# the whole tick as one captured hipGraph (sg_tick): step + terminal flags + map, outputs left in HBM
eng = sga.RolloutEngine(R, E, terminal_conditions=["max_length", "ego_collision", "ego_off_road"])
eng.upload(packed)
eng.set_road_networks([net], np.zeros(R, np.int32))
eng.step(300)
for k in range(20):
    eng.tick(act_t[k], [0, 1], torch_out=True)
t0 = time.perf_counter()
for k in range(20, 520):
    obs, fl = eng.tick(act_t[k], [0, 1], torch_out=True)
dt = (time.perf_counter() - t0) / 500
print(f"RL tick as one hipGraph launch (sg_tick: step + terminal flags + 20x20x2 map, device-resident, {R} envs): {dt*1e6:.0f} us = "
      f"{R/dt/1e6:.1f} M env-steps/s; {int((fl != 0).sum())} envs terminal")
eng.close()
