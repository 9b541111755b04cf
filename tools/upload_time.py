"""Times sg_upload (host validation + re-layout + union grids, the knots crossing PCIe beside them, device resample) for
the C3 batch, and the double-buffered pipeline: two handles, the upload of batch k + 1 on a host thread while batch k rolls out.
Usage (GPU box): python tools/upload_time.py [R] [E]      (SG_TRACE_UPLOAD=1 prints the stages of every upload)"""
import statistics
import sys
import threading
import time

sys.path.insert(0, ".")
import scenario_gym_amd as sga  # noqa: E402
from scenario_gym_amd import synthetic  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
E = int(sys.argv[2]) if len(sys.argv) > 2 else 64
T = 10000
t = time.perf_counter()
packed = synthetic.make_batch(R, E, ego_kind=sga._lib.KIND_AGENT_PID)
print(f"generate {time.perf_counter() - t:.2f} s, knots {packed.knots.nbytes / 1e9:.2f} GB")
eng = sga.RolloutEngine(R, E)
for where in ("pageable", "page-locked"):
    if where == "page-locked":
        t = time.perf_counter()
        packed.pin()  # the knots move into sg_host_alloc memory, once
        print(f"PackedScenarios.pin(): {time.perf_counter() - t:.2f} s")
    times = []
    for i in range(8):
        t = time.perf_counter()
        eng.upload(packed)
        times.append(time.perf_counter() - t)
    print(f"upload ms ({where} knots):", " ".join(f"{x * 1e3:.0f}" for x in times))
    steady = times[2:]
    print(f"sg_upload of {R} x {E} x 128 knots, {where} host memory: median {statistics.median(steady) * 1e3:.1f} ms, "
          f"min {min(steady) * 1e3:.1f} ms = {packed.knots.nbytes / min(steady) / 1e9:.1f} GB/s of knot data")
t = time.perf_counter()
eng.rollout(T)
one = time.perf_counter() - t
print(f"one rollout of {T} steps: {one * 1e3:.1f} ms; upload + rollout back to back: "
      f"{R * E * T / (statistics.median(steady) + one) / 1e9:.1f} G entity-steps/s")

# double-buffered: two handles; batch k + 1 is uploaded (host thread, ctypes releases the GIL) while batch k rolls out
engs = [eng, sga.RolloutEngine(R, E)]
engs[1].upload(packed)
n_batches = 8
t = time.perf_counter()
up = None
for k in range(n_batches):
    cur = engs[k % 2]
    if up is not None:
        up.join()
    if k + 1 < n_batches:
        up = threading.Thread(target=engs[(k + 1) % 2].upload, args=(packed,))
        up.start()
    cur.rollout(T)
dt = time.perf_counter() - t
print(f"double-buffered, {n_batches} fresh batches: {dt / n_batches * 1e3:.1f} ms per batch = "
      f"{R * E * T * n_batches / dt / 1e9:.1f} G entity-steps/s with every batch uploaded over PCIe")
for e in engs:
    e.close()
