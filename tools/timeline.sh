#!/bin/bash
# kernel timeline of one rollout: gaps between rollout-kernel launches and when the pre-pass slices ran
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
rm -rf gpurun_out/tl; timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 "$@" > gpurun_out/tl.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
ks.sort()
main = [k for k in ks if "rollout_kernel" in k[2]]
# last rollout = last 11+1 main launches
main = main[-12:]
t0 = main[0][0]
print("main launches (start, end, dur ms):")
for s, e, n in main: print(round((s - t0) / 1e6, 3), round((e - t0) / 1e6, 3), round((e - s) / 1e6, 3))
ctl = [k for k in ks if "control_kernel" in k[2] and k[0] >= t0]
print("ctl slices:", len(ctl))
import itertools
for i in range(0, len(ctl), 8):
    print(" ".join(f"{(s-t0)/1e6:.2f}-{(e-t0)/1e6:.2f}" for s, e, n in ctl[i:i+8]))
PY
