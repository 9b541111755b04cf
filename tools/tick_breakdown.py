"""Where the time of one device-resident RL tick goes (GPU box): python tools/tick_breakdown.py"""
import sys, time
sys.path.insert(0, '.')
import ctypes as C
import numpy as np
import torch
import scenario_gym_amd as sga
import scenario_gym_amd._lib as L
from scenario_gym_amd import synthetic

R, E = 4096, 64
packed = synthetic.make_batch(R, E, n_steps=2000, ego_kind=L.KIND_AGENT_VEHICLE)
rng = np.random.default_rng(1)
rings = []
for q in range(60):
    c = rng.uniform(-120, 120, 2); m = int(rng.integers(40, 120)); ang = np.sort(rng.uniform(0, 2 * np.pi, m))
    rad = rng.uniform(15, 50) * rng.uniform(0.7, 1.0, m)
    rings.append(c + rad[:, None] * np.stack([np.cos(ang), np.sin(ang)], 1))
net = dict(ring_off=np.arange(len(rings) + 1), vert_off=np.concatenate([[0], np.cumsum([len(r) for r in rings])]),
           verts=np.concatenate(rings), layers=np.ones(len(rings), np.uint32))
eng = sga.RolloutEngine(R, E); eng.upload(packed); eng.set_road_networks([net], np.zeros(R, np.int32)); eng.step(300)
acts = torch.as_tensor(synthetic.make_actions(1100, R), device="cuda:0")
lay = np.array([0, 1], np.int32); ptr = C.c_void_p()
lib, h = eng.lib, eng.h
N = 1000
T = dict(step=0.0, raster=0.0, sync=0.0, wrap=0.0)
for k in range(N + 50):
    a = acts[k:k + 1]
    t0 = time.perf_counter()
    lib.sg_step(h, 1, a.data_ptr(), 1)
    t1 = time.perf_counter()
    lib.sg_raster_map_device(h, 20.0, 20.0, 20, 20, 2, lay.ctypes.data, C.byref(ptr))
    t2 = time.perf_counter()
    lib.sg_synchronize(h)
    t3 = time.perf_counter()
    if k >= 50:
        T["step"] += t1 - t0; T["raster"] += t2 - t1; T["sync"] += t3 - t2
print({k: round(v / N * 1e6, 1) for k, v in T.items()}, "us per tick; total", round(sum(T.values()) / N * 1e6, 1))

# the same three pieces + terminal flags, separately vs as one captured graph (sg_tick), on an engine with the RL default
# terminal conditions (rollout_kernel_road)
eng.close()
eng = sga.RolloutEngine(R, E, terminal_conditions=["max_length", "ego_collision", "ego_off_road"])
eng.upload(packed); eng.set_road_networks([net], np.zeros(R, np.int32)); eng.step(300)
lib, h = eng.lib, eng.h
dfl = C.c_void_p()
for mode in ("separate", "graph"):
    t_acc = 0.0
    for k in range(N + 50):
        a = acts[k:k + 1]
        t0 = time.perf_counter()
        if mode == "separate":
            lib.sg_step(h, 1, a.data_ptr(), 1)
            lib.sg_terminal_flags(h, None, C.byref(dfl))
            lib.sg_raster_map_device(h, 20.0, 20.0, 20, 20, 2, lay.ctypes.data, C.byref(ptr))
        else:
            lib.sg_tick(h, a.data_ptr(), 1, 20.0, 20.0, 20, 20, 2, lay.ctypes.data, C.byref(ptr), C.byref(dfl))
        lib.sg_synchronize(h)
        if k >= 50:
            t_acc += time.perf_counter() - t0
    print(mode, round(t_acc / N * 1e6, 1), "us per tick (C calls + final sync)")
