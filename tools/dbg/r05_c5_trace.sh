#!/bin/bash
# kernel trace of one c5 rollout with the walker dispatch: who takes the time, chunk by chunk
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
m=${1:-4}; ch=${2:-200}
export SG_CROWD_WALK=$m SG_CROWD_CHUNK=$ch
rm -rf gpurun_out/c5trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c5trace -o t -- python3 bench.py --workload c5 --steps 1 --warmup 0 --no-cpu-baseline --verify 0 > gpurun_out/c5trace.log 2>&1
f=$(find gpurun_out/c5trace -name "*kernel_stats.csv" | head -1); cut -d, -f1-7 $f | head -8
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/c5trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("walk", "crowd"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# per chunk: the classify launch starts it
chunks, cur = [], None
for r in rows:
    n = r["Kernel_Name"]
    name = "classify" if "classify" in n else ("walk4" if "walk4" in n else ("walk" if "walk_kernel" in n else "crowd"))
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    if name == "classify":
        cur = dict(start=s, k={})
        chunks.append(cur)
    if cur is not None:
        cur["k"].setdefault(name, []).append((s, e))
        cur["end"] = e
c = chunks[min(40, len(chunks) - 1)]
print("launches of a late chunk (ms after its classify launch):", {k: [(round(s - c["start"], 2), round(e - c["start"], 2)) for s, e in v] for k, v in c["k"].items()})
for i, c in enumerate(chunks):
    if i % 5 == 0 or i == len(chunks) - 1:
        d = {k: round(sum(e - s for s, e in v), 2) for k, v in c["k"].items()}
        print(f"chunk {i:2d} at {c['start']:7.1f} ms, {c['end'] - c['start']:6.2f} ms: {d}")
PY
