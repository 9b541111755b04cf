#!/usr/bin/env python3
"""How the BASELINE config 5 crowd evolves: share of pedestrians still walking and neighbour counts over the rollout (GPU box)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scenario_gym_amd as sga
from scenario_gym_amd import synthetic

R, E, T = 64, 256, 10000
packed = synthetic.make_crowd(R, E, n_steps=T)
eng = sga.RolloutEngine(R, E, timestep=1 / 30, terminal_conditions=["max_length"], event_capacity=64)
eng.upload(packed)
done = 0
for upto in (250, 500, 1000, 1500, 2000, 3000, 5000, 7500, 10000):
    eng.step(upto - done)
    done = upto
    st = eng.state()
    xy = st["poses"][:, :, :2]
    walking = st["ctrl_state"][:, :, 1] <= 1  # goal_idx <= nwp - 1 (two-waypoint routes)
    d = np.linalg.norm(xy[:, :, None, :] - xy[:, None, :, :], axis=-1)
    nb = (d < 3.0).sum(-1) - 1
    close = (d < 1.0).sum(-1) - 1
    coll = np.array([bin(int(w)).count("1") for w in st["coll"].ravel()]).reshape(R, E, -1).sum(-1)
    print(f"step {upto:6d}: walking {walking.mean():.3f}  neighbours(3m) mean {nb.mean():.1f} max {nb.max()}  of walkers {nb[walking].mean() if walking.any() else 0:.1f}"
          f"  within 1m {close.mean():.2f}  collisions/entity {coll.mean():.2f}")
eng.close()
