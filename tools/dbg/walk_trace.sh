#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/wtrace -o t -- python3 tools/dbg/walk_pmc.py > gpurun_out/wtrace.log 2>&1
f=$(find gpurun_out/wtrace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
ks = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Kernel_Name"]) for r in rows if "spin" not in r["Kernel_Name"]]
ks.sort()
def short(n):
    for k in ("walk_kernel<1>", "walk_kernel<2>", "walk_classify", "rollout_kernel_crowd", "build_grid"):
        if k in n: return k
    return n[:30]
# per chunk: print start/end of each kernel for a few chunks
out = []
for s, e, n in ks: out.append((s / 1e6, e / 1e6, (e - s) / 1e6, short(n)))
cls = [i for i, o in enumerate(out) if o[3] == "walk_classify"]
for ci in cls[::5]:
    seg = out[ci:ci + 5]
    print(" | ".join(f"{o[3][:14]} {o[0]:.2f}-{o[1]:.2f} ({o[2]:.2f})" for o in seg))
tot = collections.defaultdict(float)
for o in out: tot[o[3]] += o[2]
print({k: round(v, 1) for k, v in tot.items()}, "wall", round(out[-1][1] - out[0][0], 1))
PY
rm -rf gpurun_out/wtrace
