#!/bin/bash
# Usage (GPU box): [SGYM_LIB=...] bash tools/rss_pmc.sh   -- SQ counters of rollout_kernel_rss per wavefront-step (tools/rss_time.py: 4096 x 64 x 2000)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
rm -rf gpurun_out/rss_pmc gpurun_out/rss_pmc2
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM --output-format csv -d gpurun_out/rss_pmc -o p -- python3 tools/rss_time.py > gpurun_out/rss_pmc.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_IFETCH --output-format csv -d gpurun_out/rss_pmc2 -o p -- python3 tools/rss_time.py > gpurun_out/rss_pmc2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
out = {}
for d in ("rss_pmc", "rss_pmc2"):
    fs = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no csv"); continue
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        if "rollout_kernel_rss" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in sorted(agg.items()): out[k] = round(v / (4096 * 4000), 1)  # two rollouts of 2000 steps
print("rollout_kernel_rss per wavefront-step:", out)
PY
