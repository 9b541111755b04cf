#!/usr/bin/env python3
"""Experiment (GPU box): does pairing a crowd workgroup in its heavy phase (steps ~1,000-4,000: everybody crosses the square)
with one in its light phase (the late steps: ~31 walkers of 256) on the same compute unit shorten config 5?

The 1024 x 256 batch as four handles of 256 scenarios (one workgroup per compute unit each), two host threads:
  plain:     thread X: H0 full, H1 full          thread Y: H2 full, H3 full            (both residents of a unit in the same phase)
  staggered: thread X: H0 full, H1 full          thread Y: H3 first half, H2 full, H3 second half   (half a rollout apart)
Prints the wall time of both schedules, alternating, three repetitions.  No product code involved beyond sg_rollout_async."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

import scenario_gym_amd as sga
from scenario_gym_amd import synthetic

R, E, T = 1024, 256, 10000
packed = synthetic.make_crowd(R, E, n_steps=T)
H = []
for k in range(4):
    eng = sga.RolloutEngine(R // 4, E, timestep=1 / 30, terminal_conditions=["max_length"], event_capacity=64)
    eng.upload(packed.shard(k * R // 4, (k + 1) * R // 4))
    eng.rollout(64)
    H.append(eng)


def run(seq):
    for eng, n, reset in seq:
        eng.rollout_async(n, reset)
        eng.synchronize()


def timed(x, y):
    tx, ty = threading.Thread(target=run, args=(x,)), threading.Thread(target=run, args=(y,))
    t0 = time.perf_counter()
    tx.start(); ty.start(); tx.join(); ty.join()
    return time.perf_counter() - t0


def digest():
    return [float(np.nansum(h.state()["poses"])) for h in H]


plain = ([(H[0], T, True), (H[1], T, True)], [(H[2], T, True), (H[3], T, True)])
for split in (T // 2, 3000, 7000):
    stag = ([(H[0], T, True), (H[1], T, True)], [(H[3], split, True), (H[2], T, True), (H[3], T - split, False)])
    for rep in range(3):
        a = timed(*plain)
        da = digest()
        b = timed(*stag)
        db = digest()
        print(f"split {split}: plain {a * 1e3:.1f} ms ({R * E * T / a / 1e9:.2f} G)   staggered {b * 1e3:.1f} ms ({R * E * T / b / 1e9:.2f} G)   same state: {da == db}", flush=True)
