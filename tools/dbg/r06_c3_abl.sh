#!/bin/bash
# c3 with phases compiled out (tools/experiments/r06_c3_ablations.patch; wrong results on purpose): time and SQ instruction counts per wavefront-step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for v in full NO_COLL NO_FILTER NO_STORES NO_STATS NO_EVENTS; do
  L=""; [ "$v" != "full" ] && export SGYM_LIB=scenario_gym_amd/lib/ab/abl_$v.so || unset SGYM_LIB
  timeout 300 python3 bench.py --no-cpu-baseline --no-configs --verify 0 --steps 6 --warmup 2 > gpurun_out/r06_abl_$v.json 2> gpurun_out/r06_abl_$v.err
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/r06_abl_pmc_$v -o p -- python3 bench.py --no-cpu-baseline --no-configs --verify 0 --steps 1 --warmup 0 > gpurun_out/r06_abl_pmc_$v.log 2>&1
  python3 - $v <<'PY'
import csv, glob, json, sys, collections
v = sys.argv[1]
l = json.load(open(f"gpurun_out/r06_abl_{v}.json"))
agg = collections.defaultdict(float)
for f in glob.glob(f"gpurun_out/r06_abl_pmc_{v}/**/*counter_collection.csv", recursive=True)[:1]:
    for r in csv.DictReader(open(f)):
        if "rollout_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
ws = 4096 * 10000
print(f"{v:10s} {l['value'] / 1e9:7.2f} G  kernel {l['roofline']['kernel_ms']:6.2f} ms  per wavefront-step:", {k: round(x / ws, 1) for k, x in sorted(agg.items())})
PY
done
