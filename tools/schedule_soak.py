#!/usr/bin/env python3
"""Soak test of the launch schedule: ONE process with torch imported creates, uses and destroys N handles
of mixed shapes -- table-path batches of several tile widths and sizes, a time-sliced batch, a crowd, an RSS batch, a hipGraph
tick, pinned uploads through sg_group -- and logs, for every handle, which schedule its rollout ran (sg_schedule_info) and
whether the result equals the first handle of the same shape bit for bit.  Rounds 3-4 decided the schedule with a timing probe
at sg_create, and a long-lived process could get one pipeline of three; the persistent launch has nothing to probe: every
table-path rollout here must report schedule 2 ("persistent_queue"), launches 1.

    python tools/schedule_soak.py [N=300] > profiles/r05_schedule_soak.txt
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (the driver's pytest process has it imported, with its streams)

import scenario_gym_amd as sga
import scenario_gym_amd._lib as L
from scenario_gym_amd import synthetic


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    torch.zeros(8, device="cuda:0")  # torch's context and streams exist first
    shapes = [  # (name, R, E, steps, ego kind, expected schedule)
        ("tab64", 203, 64, 200, L.KIND_AGENT_PID, 2), ("tab24", 500, 24, 150, L.KIND_AGENT_VEHICLE, 2),
        ("tab12", 64, 12, 120, L.KIND_AGENT_PID, 2), ("big", 4096, 64, 300, L.KIND_AGENT_PID, 2),
        ("sliced", 256, 16, 600, None, 0), ("tick", 48, 12, 40, L.KIND_AGENT_VEHICLE, None),
    ]
    packed = {}
    for name, R, E, steps, kind, _ in shapes:
        packed[name] = synthetic.make_batch(R, E, n_steps=steps, ego_kind=kind, extent=25.0 if E > 12 else 10.0) if kind is not None \
            else synthetic.make_batch(R, E, n_steps=steps, extent=12.0)
    first = {}
    counts = {}
    bad = 0
    t0 = time.time()
    for i in range(n):
        name, R, E, steps, kind, want = shapes[i % len(shapes)]
        eng = sga.RolloutEngine(R, E, event_capacity=32)
        eng.upload(packed[name])
        if name == "tick":  # the graph-captured RL tick: one step per call
            acts = synthetic.make_actions(steps, R)
            for k in range(steps):
                eng.step(1, acts[k:k + 1])
            info = dict(schedule=None, launches=None)
        else:
            eng.rollout(steps)
            if i % 12 == 3:  # ... continued in pieces now and then
                eng.rollout_async(17, do_reset=False)
                eng.synchronize()
            info = eng.schedule_info()
        rows, ev = eng.metrics()
        st = eng.state()
        key = (rows.tobytes(), ev.tobytes(), st["poses"].tobytes())
        same = first.setdefault((name, i % 12 == 3), key) == key
        ok = same and (want is None or (info["schedule"] == want and (want != 2 or info["launches"] == 1)))
        bad += not ok
        counts[(name, info["schedule"])] = counts.get((name, info["schedule"]), 0) + 1
        print(f"create {i:3d} {name:6s} {R:4d} x {E:2d} x {steps:3d}: schedule {info['schedule']} launches {info['launches']} "
              f"same-as-first {same}{'' if ok else '   <-- UNEXPECTED'}", flush=True)
        eng.close()
    print(f"{n} handles in {time.time() - t0:.1f} s; (shape, schedule) counts: {sorted(counts.items(), key=str)}; unexpected: {bad}")
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
