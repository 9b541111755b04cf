#!/bin/bash
# second SQ counter set (waits by class, scalar / LDS / VMEM-write activity, branches); same conventions as pmc_sq.sh
tag=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/${tag}_pmc_sq2 -o p -- python3 bench.py --steps 1 --warmup 0 --sim-steps 2000 --no-cpu-baseline "$@" > gpurun_out/${tag}_pmc_sq2.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/{tag}_pmc_sq2/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if "rollout_kernel" in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
print({k: round(v / (4096 * 2000), 1) for k, v in agg.items()})
PY
