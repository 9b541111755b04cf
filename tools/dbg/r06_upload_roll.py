#!/usr/bin/env python3
"""sg_upload followed by sg_rollout on the c3 batch, several times: how long the rollout right after an upload takes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import scenario_gym_amd as sga
from scenario_gym_amd import synthetic
R, E, T = 4096, 64, 10000
packed = synthetic.make_batch(R, E, ego_kind=sga._lib.KIND_AGENT_PID)
packed.pin()
eng = sga.RolloutEngine(R, E)
out = []
for i in range(5):
    t = time.perf_counter(); eng.upload(packed); u = time.perf_counter() - t
    t = time.perf_counter(); eng.rollout(T); a = time.perf_counter() - t
    t = time.perf_counter(); eng.rollout(T); b = time.perf_counter() - t
    out.append((round(u * 1e3, 1), round(a * 1e3, 1), round(b * 1e3, 1), eng.schedule_info()["schedule"]))
print(os.environ.get("SGYM_LIB", "product"), "upload / first rollout / second rollout ms, schedule:", out)
eng.close()
