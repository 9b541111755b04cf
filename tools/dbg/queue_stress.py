"""Stress of the persistent launch on small batches: first call on a fresh handle vs the second vs chunk launches (SG_QUEUE=0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scenario_gym_amd as sga
import scenario_gym_amd._lib as L
from scenario_gym_amd import synthetic

rng = np.random.default_rng(int(os.environ.get("QS_SEED", "1")))
N = int(os.environ.get("QS_N", "150"))
bad = 0
t0 = time.time()
junk = []
for it in range(N):
    E = int(rng.choice([4, 8, 16, 16, 30, 64]))
    R = int(rng.integers(8, 80))
    steps = int(rng.integers(200, 700))
    ego = [L.KIND_AGENT_PID, L.KIND_AGENT_VEHICLE][int(rng.integers(0, 2))]
    term = [["max_length"], ["max_length", "collision"], ["max_length", "ego_collision"]][int(rng.integers(0, 3))]
    dt = float(rng.choice([1 / 30, 0.1]))
    packed = synthetic.make_batch(R, E, n_steps=steps, timestep=dt, ego_kind=ego, static_frac=0.15, vanish_frac=0.25,
                                  extent=30.0 if E > 16 else 14.0, seed=int(rng.integers(1, 1 << 30)))
    # dirty the allocator: a few buffers of random size filled with noise, freed before the engines allocate
    import torch
    for _ in range(3):
        junk.append(torch.full((int(rng.integers(1, 64)) << 18,), float("nan"), dtype=torch.float64, device="cuda:0"))
    junk = junk[-2:]
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    res = []
    for q in ("1", "1", "0"):
        os.environ["SG_QUEUE"] = q
        eng = sga.RolloutEngine(R, E, timestep=dt, terminal_conditions=term, event_capacity=8)
        eng.set_slicing(False)
        eng.upload(packed)
        outs = []
        for rep in range(2 if q == "1" else 1):
            eng.rollout(steps)
            st = eng.state()
            rows, ev = eng.metrics()
            outs.append((st, rows.copy(), ev.copy(), eng.schedule_info()["schedule"]))
        eng.close()
        res.append(outs)
    ref = res[2][0]
    for name, o in (("fresh handle, call 1", res[0][0]), ("fresh handle, call 2", res[0][1]), ("second handle, call 1", res[1][0])):
        for f in ("poses", "vels", "dists", "ctrl_state", "present", "coll", "t", "n_steps", "done"):
            a, b = np.asarray(o[0][f]), np.asarray(ref[0][f])
            if a.tobytes() != b.tobytes():
                idx = np.argwhere(~((a == b) | ((a != a) & (b != b))))[:4]
                print(f"MISMATCH it={it} R={R} E={E} steps={steps} ego={ego} term={term} dt={dt:.3f} [{name}] {f} at {idx.tolist()} "
                      f"got {[a[tuple(i)] for i in idx]} want {[b[tuple(i)] for i in idx]} n_steps {o[1]['n_steps'][idx[:, 0]].tolist()} schedule {o[3]}", flush=True)
                bad += 1
                break
        if o[1].tobytes() != ref[1].tobytes() or o[2].tobytes() != ref[2].tobytes():
            print(f"MISMATCH it={it} [{name}] metric rows / events", flush=True)
            bad += 1
print(f"{N} batches, {bad} mismatches, {time.time() - t0:.0f} s")
