import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import scenario_gym_amd as sga
from scenario_gym_amd import synthetic
R, E, T = int(os.environ.get("R", "1024")), 256, int(os.environ.get("T", "10000"))
packed = synthetic.make_crowd(R, E, n_steps=T)
eng = sga.RolloutEngine(R, E, terminal_conditions=["max_length"], event_capacity=64)
eng.upload(packed)
eng.rollout(T)
print("STATS", json.dumps(dict(eng.crowd_walk_stats(), kernel_ms=eng.last_kernel_ms(), chunk=int(os.environ.get("SG_CROWD_CHUNK", "200")))))
eng.close()
