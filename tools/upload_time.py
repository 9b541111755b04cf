"""Times sg_upload (host validation + re-layout + union grids + PCIe + device resample) for the C3 batch.
Usage (GPU box): python tools/upload_time.py [R] [E]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import scenario_gym_amd as sga  # noqa: E402
from scenario_gym_amd import synthetic  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
E = int(sys.argv[2]) if len(sys.argv) > 2 else 64
t = time.perf_counter()
packed = synthetic.make_batch(R, E, ego_kind=sga._lib.KIND_AGENT_PID)
print(f"generate {time.perf_counter() - t:.2f} s, knots {packed.knots.nbytes / 1e9:.2f} GB")
eng = sga.RolloutEngine(R, E)
for i in range(3):
    t = time.perf_counter()
    eng.upload(packed)
    dt = time.perf_counter() - t
    print(f"upload {i}: {dt * 1e3:.0f} ms = {packed.knots.nbytes / dt / 1e9:.2f} GB/s of knot data")
eng.close()
