#!/bin/bash
# Kernel timeline of ONE rollout of a bench workload (default c3): every rollout-kernel / pre-pass launch of the last rollout with
# its hardware queue, start, end and duration (ms after the rollout's reset launch) -- the pipelines of DESIGN 3.0 side by side.
# Usage (GPU box): bash tools/timeline.sh [bench args] > gpurun_out/<tag>_timeline.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
rm -rf gpurun_out/tl; timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python3 bench.py --no-cpu-baseline --verify 0 --steps 1 --warmup 1 "$@" > gpurun_out/tl.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if any(k in r["Kernel_Name"] for k in ("rollout_kernel", "control_kernel", "rss_lines"))]
resets = [i for i, r in enumerate(sel) if "rollout_kernel<" in r["Kernel_Name"] or ("rollout_kernel_rss<" in r["Kernel_Name"])]
i0 = resets[-1] if resets else 0
t0 = int(sel[i0]["Start_Timestamp"])
print("kernel                              queue   start      end      dur (ms)")
busy = []
for r in sel[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("sg::", "").split("(")[0]
    print(f"{name:36s} {r['Queue_Id']:>3s} {(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f} {(e - s) / 1e6:8.3f}")
    if "rollout_kernel" in r["Kernel_Name"]:
        busy.append((s, e))
busy.sort()
u, lo, hi = 0, None, None
for s, e in busy:
    if hi is None or s > hi:
        if hi is not None: u += hi - lo
        lo, hi = s, e
    else:
        hi = max(hi, e)
if hi is not None: u += hi - lo
print(f"rollout-kernel launches: {len(busy)}, sum of durations {sum(e - s for s, e in busy) / 1e6:.3f} ms, union of their intervals {u / 1e6:.3f} ms, "
      f"first start to last end {(max(e for s, e in busy) - min(s for s, e in busy)) / 1e6:.3f} ms")
PY
