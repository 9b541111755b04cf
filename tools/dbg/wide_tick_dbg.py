import numpy as np, sys
sys.path.insert(0, "/root/repo")
import scenario_gym_amd as sga
import scenario_gym_amd._lib as L
from scenario_gym_amd import synthetic
R, E = 3, 700
packed = synthetic.make_batch(R, E, n_steps=60, ego_kind=L.KIND_AGENT_VEHICLE, extent=70.0, vanish_frac=0.2)
acts = synthetic.make_actions(40, R)
sq = np.array([[-9.0, -9.0], [12.0, -9.0], [12.0, 10.0], [-9.0, 10.0]])
net = dict(ring_off=[0, 1], vert_off=[0, 4], verts=sq, layers=[1 | 2])
a, b = (sga.RolloutEngine(R, E, terminal_conditions=["max_length", "ego_collision"]) for _ in range(2))
for e in (a, b):
    e.upload(packed)
    e.set_road_networks([net], np.array([0, -1, 0], np.int32))
geo = dict(layers=[0, 1], width=30.0, height=24.0, nw=20, nh=16)
for k in range(3):
    a.step(1, acts[k:k + 1])
    m1 = a.raster_map(geo["layers"], geo["width"], geo["height"], geo["nw"], geo["nh"])
    m2 = a.raster_map(geo["layers"], geo["width"], geo["height"], geo["nw"], geo["nh"])
    obs, fl = b.tick(acts[k], **geo)
    m3 = b.raster_map(geo["layers"], geo["width"], geo["height"], geo["nw"], geo["nh"])
    print(k, "a twice equal", np.array_equal(m1, m2), "tick vs a", [(r, l, int((obs[r, l] != m1[r, l]).sum())) for r in range(R) for l in range(2)],
          "b.raster vs a", np.array_equal(m3, m1), "sums", obs.sum(axis=(2, 3)).tolist(), m1.sum(axis=(2, 3)).tolist())
