#!/bin/bash
# SQ_INSTS_VALU / SALU per wave-step of the c3 headline kernel for the library given in SGYM_LIB (ablation builds)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for lib in ${LIBS:-"" scenario_gym_amd/lib/ab/nocoll.so}; do
  tag=$(basename "${lib:-base}" .so)
  SGYM_LIB=$lib timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmc_$tag -o p -- python3 bench.py --no-cpu-baseline --verify 0 --steps 1 --warmup 0 > gpurun_out/pmc_$tag.log 2>&1
  python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(float)
for f in glob.glob(f"gpurun_out/pmc_{tag}/**/*counter_collection.csv", recursive=True)[:1]:
    for r in csv.DictReader(open(f)):
        if "rollout_kernel_tab" in r["Kernel_Name"]: agg[r["Counter_Name"]] += float(r["Counter_Value"])
ws = 4096 * 10000
print(tag, {k: round(v / ws, 1) for k, v in agg.items()})
PY
  rm -rf gpurun_out/pmc_$tag
done
