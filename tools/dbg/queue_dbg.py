"""Debug run of the persistent table launch on a small batch: the queue words watched from the host (SG_QUEUE_DEBUG)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("SG_QUEUE_DEBUG", "60")
os.environ.setdefault("SG_QUEUE_TIMEOUT_MS", "1000")
import numpy as np
import scenario_gym_amd as sga
import scenario_gym_amd._lib as L
from scenario_gym_amd import synthetic

R, E, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 203, 64, int(sys.argv[2]) if len(sys.argv) > 2 else 150
packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=L.KIND_AGENT_PID, extent=25.0)
eng = sga.RolloutEngine(R, E, terminal_conditions=["max_length"], event_capacity=64)
eng.set_tuning(tab_min_steps=1, chunk_steps=int(sys.argv[3]) if len(sys.argv) > 3 else 16)
eng.upload(packed)
t0 = time.time()
try:
    eng.rollout(steps)
    print("rollout ok", time.time() - t0, eng.schedule_info(), flush=True)
    rows, ev = eng.metrics()
    print("n_steps", rows["n_steps"][:8], flush=True)
except Exception as e:
    print("FAILED", e, flush=True)
eng.close()
