"""Stage timings of sg_upload for the C3 batch, page-locked knots (SG_TRACE_UPLOAD=1): python tools/upload_stages.py"""
import os, sys, time
os.environ["SG_TRACE_UPLOAD"] = "1"
sys.path.insert(0, ".")
import scenario_gym_amd as sga
from scenario_gym_amd import synthetic
R, E = 4096, 64
packed = synthetic.make_batch(R, E, ego_kind=sga._lib.KIND_AGENT_PID)
if len(sys.argv) < 2 or sys.argv[1] != "pageable":
    packed.pin()
eng = sga.RolloutEngine(R, E)
for i in range(12):
    t = time.perf_counter()
    eng.upload(packed)
    print(f"--- upload {i}: {(time.perf_counter() - t) * 1e3:.1f} ms", file=sys.stderr, flush=True)
eng.close()
