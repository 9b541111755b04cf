#!/bin/bash
# Usage (on the GPU box): bash tools/profile_round.sh <tag>
# Produces under gpurun_out/: <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `python3 bench.py`),
# <tag>_pmc_sq.txt (SQ instruction mix per wave-step) and <tag>_hbm_traffic.json (FETCH_SIZE / WRITE_SIZE passes).
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_bench_under_trace.log 2>&1
f=$(find gpurun_out/${tag}_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${tag}_kernel_stats.csv
tail -1 gpurun_out/${tag}_bench_under_trace.log | cut -c1-400
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/${tag}_pmc_$c -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/${tag}_pmc_$c.log 2>&1
done
python3 - "$tag" <<'PY'
import csv, glob, json, sys
tag = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/{tag}_pmc_{c}/**/*counter_collection.csv", recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if "rollout_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
    by = {}
    for r in rows:
        by[r["Dispatch_Id"]] = by.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    out[c] = max(by.values())  # the full-length rollout launch (the reset-only launch is tiny)
rec = dict(scenarios=4096, entities=64, sim_steps=10000, fetch_size_kb=out["FETCH_SIZE"], write_size_kb=out["WRITE_SIZE"],
           hbm_bytes_per_launch=(out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024,
           note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, KB units x1024; the kernel's loads are 8 B/lane, "
                "so the x2 FETCH_SIZE correction calibrated for 16-B/lane streams (MI355X_MICROARCH.md, HBM) is NOT applied")
json.dump(rec, open(f"gpurun_out/{tag}_hbm_traffic.json", "w"), indent=1)
print(rec)
PY
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/${tag}_pmc_sq -o p -- python3 bench.py --steps 1 --warmup 0 --sim-steps 2000 --no-cpu-baseline > gpurun_out/${tag}_pmc_sq.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/{tag}_pmc_sq/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if "rollout_kernel" in r["Kernel_Name"]:
        agg[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
d = sorted({k[0] for k in agg}, key=int)[-1]
ws = 4096 * 2000
line = {k[1]: round(v / ws, 1) for k, v in agg.items() if k[0] == d}
open(f"gpurun_out/{tag}_pmc_sq.txt", "w").write("per wave-step (4096 waves x 2000 steps; *_CYCLES and ACTIVE/WAIT in quad-cycles): %s\n" % line)
print(line)
PY
