cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc_c5 -o p -- python3 bench.py --workload c5 --steps 1 --warmup 0 --sim-steps 300 --no-cpu-baseline > gpurun_out/pmc_c5.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_c5/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if "rollout_kernel" in r["Kernel_Name"]:
        agg[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
d = sorted({k[0] for k in agg}, key=int)[-1]
ws = 4096 * 300
print({k[1]: round(v / ws, 1) for k, v in agg.items() if k[0] == d})
PY
tail -1 gpurun_out/pmc_c5.log | cut -c1-200
