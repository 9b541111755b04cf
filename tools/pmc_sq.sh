#!/bin/bash
# Usage (GPU box): bash tools/pmc_sq.sh <tag> [bench args...]   -> gpurun_out/<tag>_pmc_sq.txt
# SQ instruction mix of the rollout kernel per wave-step, summed over all its dispatches of ONE rollout of the workload (all its steps:
# the crowd's instruction mix changes over the rollout, bench.py multiplies these by the wave-steps / s of the same rollout).
tag=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM --output-format csv -d gpurun_out/${tag}_pmc_sq -o p -- python3 bench.py --no-cpu-baseline --no-configs "$@" --steps 1 --warmup 0 > gpurun_out/${tag}_pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d gpurun_out/${tag}_pmc_sq_b -o p -- python3 bench.py --no-cpu-baseline --no-configs "$@" --steps 1 --warmup 0 > gpurun_out/${tag}_pmc_sq_b.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
line = json.loads([l for l in open(f"gpurun_out/{tag}_pmc_sq.log") if l.startswith("{")][-1])
cfg = line["config"]
E = cfg["entities"]
EP = max(4, 1 << (E - 1).bit_length()) if E <= 64 else (128 if E <= 128 else 256)  # entity stride (tile lanes)
waves = -(-cfg["scenarios_per_gpu"] * EP // 64)
agg = collections.defaultdict(float)
for f in glob.glob(f"gpurun_out/{tag}_pmc_sq/**/*counter_collection.csv", recursive=True)[:1] + \
        glob.glob(f"gpurun_out/{tag}_pmc_sq_b/**/*counter_collection.csv", recursive=True)[:1]:
    for r in csv.DictReader(open(f)):
        k = "rollout" if "rollout_kernel" in r["Kernel_Name"] else ("control" if "control_kernel" in r["Kernel_Name"] else None)
        if k:
            agg[(k, r["Counter_Name"])] += float(r["Counter_Value"])
T = cfg["sim_steps"]
ws = waves * T
out = {k[1]: round(v / ws, 1) for k, v in agg.items() if k[0] == "rollout"}
ctl = {k[1]: round(v / T, 1) for k, v in agg.items() if k[0] == "control"}
txt = "rollout_kernel per wave-step (%d waves x %d steps; *_CYCLES / ACTIVE / WAIT in quad-cycles): %s\ncontrol_kernel per step, all waves: %s\n" % (waves, T, out, ctl)
open(f"gpurun_out/{tag}_pmc_sq.txt", "w").write(txt)
wl = cfg["name"]
rec = dict(scenarios=cfg["scenarios_per_gpu"], entities=E, sim_steps=T, src_sha16=line["roofline"]["src_sha16"],
           kernel=line["roofline"]["kernel"], waves=waves, per_wave_step=out, control_kernel_per_step=ctl,
           note="rocprofv3 --pmc (SQ counters only) over one rollout of the workload, summed over the rollout-kernel dispatches and "
                "divided by wavefronts x steps; *_CYCLES / ACTIVE / WAIT in quad-cycles.  The persistent table launch "
                "(rollout_kernel_tabq*) carries the controller pre-pass as a role of the same kernel: its instructions (~6.6 VALU per "
                "wavefront-step of the 4096 x 64 batch) and the cycles its wavefronts and the waiting rollout wavefronts are "
                "resident are in these figures")
json.dump(rec, open(f"gpurun_out/{tag}_pmc_sq.json", "w"), indent=1)
json.dump(rec, open(f"gpurun_out/latest_{wl}_pmc_sq.json", "w"), indent=1)
print(txt)
PY
