"""Where the time of the persistent table launch goes on the c3 shape: per-item time stamps (SG_QUEUE_TIMES) -> summary."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scenario_gym_amd as sga
import scenario_gym_amd._lib as L
from scenario_gym_amd import synthetic

R, E, steps = int(os.environ.get("QT_R", 4096)), 64, 10000
packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=L.KIND_AGENT_PID)
eng = sga.RolloutEngine(R, E, terminal_conditions=["max_length"], event_capacity=64)
eng.set_slicing(False)
if os.environ.get("QT_CHUNK"):
    eng.set_tuning(chunk_steps=int(os.environ["QT_CHUNK"]))
eng.upload(packed)
eng.rollout(steps)
eng.rollout(steps)
print("untraced kernel ms", eng.last_kernel_ms())
path = "/tmp/qtimes.bin"
os.environ["SG_QUEUE_TIMES"] = path
eng.rollout(steps)
print("traced kernel ms", eng.last_launch_stats(), eng.schedule_info())
eng.close()
blob = open(path, "rb").read()
raw = np.frombuffer(blob[:len(blob) // 8 * 8], dtype=np.uint64)
C, nblk, ncw, grid = (int(x) for x in raw[:4])
n_items = C * nblk
t = raw[4:4 + n_items * 4].reshape(C, nblk, 4).astype(np.float64)
ctl = raw[4 + n_items * 4:4 + n_items * 4 + ncw * C].reshape(ncw, C).astype(np.float64)
k0 = np.frombuffer(blob[(4 + n_items * 4 + ncw * C) * 8:], dtype=np.int32)[:C + 1]
t0 = t[..., 0].min()
t, ctl = (t - t0) / 100.0, (ctl - t0) / 100.0  # us
print(f"C {C} nblk {nblk} pre-pass waves {ncw} grid {grid}; chunk lengths {np.diff(k0).tolist()}")
print(f"end of last item {t[..., 3].max() / 1e3:.2f} ms")
wait, body, hand = t[..., 1] - t[..., 0], t[..., 2] - t[..., 1], t[..., 3] - t[..., 2]
slots = grid - ncw
print(f"sum over items / slots: wait {wait.sum() / slots / 1e3:.2f} ms, body {body.sum() / slots / 1e3:.2f} ms, hand-off {hand.sum() / slots / 1e3:.2f} ms")
print("chunk: len | pre-pass done at (ms, min..max over its wavefronts) | first pull .. last done (ms) | body us/step p10 p50 p90 | wait us p50 p99 | hand-off us p50")
for c in range(C):
    n = k0[c + 1] - k0[c]
    b = body[c] / n
    print(f"{c:2d}: {n:5d} | {ctl[:, c].min() / 1e3:6.2f} .. {ctl[:, c].max() / 1e3:6.2f} | {t[c, :, 0].min() / 1e3:6.2f} .. {t[c, :, 3].max() / 1e3:6.2f} | "
          f"{np.percentile(b, 10):.2f} {np.percentile(b, 50):.2f} {np.percentile(b, 90):.2f} | {np.percentile(wait[c], 50):7.1f} {np.percentile(wait[c], 99):8.1f} | {np.percentile(hand[c], 50):.1f}")
# how busy the slots are over time: items in their body per 1 ms bucket
end = t[..., 3].max()
print("pre-pass wavefronts, ms at which each finished its last chunk:", np.round(np.sort(ctl[:, -1]) / 1e3, 1).tolist())
for lo in np.arange(0, end, 1000.0):
    inb = ((t[..., 1] < lo + 1000) & (t[..., 2] > lo + 1000)).sum()
    inw = ((t[..., 0] < lo + 1000) & (t[..., 1] > lo + 1000)).sum()
    print(f"t = {lo / 1e3 + 1:5.1f} ms: {inb} items in their body, {inw} waiting (of {slots} rollout wavefronts)")
# the chain of every block: its items one after the other -- is the call bound by the slots (work / slots) or by its slowest blocks?
per_block_body = body.sum(axis=0) / 1e3
finish = t[-1, :, 3] / 1e3
gap = (t[1:, :, 1] - t[:-1, :, 3]).sum(axis=0) / 1e3  # between the end of an item and the start of the block's next one
q = lambda a, f: float(np.percentile(a, f))
print(f"per block, ms: body of all its items p50 {q(per_block_body, 50):.2f} p90 {q(per_block_body, 90):.2f} p99 {q(per_block_body, 99):.2f} max {per_block_body.max():.2f} | "
      f"between its items p50 {q(gap, 50):.2f} p99 {q(gap, 99):.2f} | finished at p50 {q(finish, 50):.2f} p90 {q(finish, 90):.2f} max {finish.max():.2f}")
slow = np.argsort(per_block_body)[-5:]
print("the five slowest blocks (body ms, gaps ms, finished at):", [(int(b), round(float(per_block_body[b]), 2), round(float(gap[b]), 2), round(float(finish[b]), 2)) for b in slow])
print(f"work / slots: {body.sum() / slots / 1e3:.2f} ms")
