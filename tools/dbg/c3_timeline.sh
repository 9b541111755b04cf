#!/bin/bash
# kernel timeline of one c3 pass (base and the no-collision ablation): who waits for whom
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for lib in ${LIBS:-"" scenario_gym_amd/lib/ab/nocoll.so}; do
  tag=$(basename "${lib:-base}" .so)
  SGYM_LIB=$lib timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$tag -o t -- python3 bench.py --no-cpu-baseline --verify 0 --steps 1 --warmup 1 > gpurun_out/tl_$tag.log 2>&1
  python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
rows = []
for f in glob.glob(f"gpurun_out/tl_{tag}/**/*kernel_trace.csv", recursive=True)[:1]:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "rollout_kernel" in n or "control_kernel" in n or "event_ego" in n:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "tab" if "rollout" in n else ("ctl" if "control" in n else "ego"), r.get("Queue_Id", "?"), r.get("Grid_Size", "?")))
rows.sort()
# last pass: the last 1/2 of the tab launches
tabs = [r for r in rows if r[2] == "tab"]
n = len(tabs) // 2
t0 = tabs[-n][0]
last = [r for r in rows if r[0] >= t0 - 3_000_000]
base = last[0][0]
print(tag, "launches in the last pass:", len(last), "span ms", (max(r[1] for r in last) - base) / 1e6)
for r in last[:60]:
    print(f"  {r[2]} q{r[3]} grid {r[4]:>8} start {(r[0]-base)/1e6:8.3f} dur {(r[1]-r[0])/1e6:7.3f}")
PY
done
