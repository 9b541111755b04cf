#!/bin/bash
# Usage (on the GPU box): bash tools/profile_round.sh <tag> [bench args, e.g. --workload c5 --steps 1]
# Produces under gpurun_out/ (copy what is to be judged into profiles/):
#   <tag>_bench.json         the plain bench line (no profiler attached)
#   <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the same command
#   <tag>_hbm_traffic.json   FETCH_SIZE / WRITE_SIZE (separate --pmc passes), summed over the rollout-kernel
#                            launches of ONE rollout (a long rollout is cut into chunks of steps)
#   <tag>_pmc_sq.txt/.json   SQ instruction mix per wave-step (tools/pmc_sq.sh)
#   latest_<workload>_{hbm_traffic,pmc_sq}.json   the same two records under the names bench.py reads from profiles/
tag=${1:-r01}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
python3 bench.py "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
cut -c1-300 gpurun_out/${tag}_bench.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -o t -- python3 bench.py --no-cpu-baseline --no-configs "$@" > gpurun_out/${tag}_bench_under_trace.log 2>&1
f=$(find gpurun_out/${tag}_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${tag}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d gpurun_out/${tag}_pmc_$c -o p -- python3 bench.py --no-cpu-baseline --no-configs "$@" --steps 1 --warmup 0 > gpurun_out/${tag}_pmc_$c.log 2>&1
done
python3 - "$tag" <<'PY'
import csv, glob, json, sys
tag = sys.argv[1]
line = json.loads([l for l in open(f"gpurun_out/{tag}_pmc_WRITE_SIZE.log") if l.startswith("{")][-1])
cfg = line["config"]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/{tag}_pmc_{c}/**/*counter_collection.csv", recursive=True)
    out[c] = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(f[0]))
                 if "rollout_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c)
wl = cfg["name"]
# calibration on this library's access shape (512-byte rows, 8 B per lane; tools/hbm_calib.sh -> profiles/r03_counter_calibration.json):
# FETCH_SIZE reports half of the bytes read (factor 2.0), WRITE_SIZE all of them (factor 1.0)
try:
    cal = json.load(open("profiles/r03_counter_calibration.json"))
    ff, wf = round(cal["fetch_factor"], 3), round(cal["write_factor"], 3)
except Exception:
    ff, wf = 2.0, 1.0
rec = dict(scenarios=cfg["scenarios_per_gpu"], entities=cfg["entities"], sim_steps=cfg["sim_steps"],
           src_sha16=line["roofline"]["src_sha16"], kernel=line["roofline"]["kernel"],
           fetch_size_kb=out["FETCH_SIZE"], write_size_kb=out["WRITE_SIZE"], fetch_factor=ff, write_factor=wf,
           hbm_bytes_per_rollout=(out["FETCH_SIZE"] * ff + out["WRITE_SIZE"] * wf) * 1024,
           launches_per_rollout=line["roofline"]["launches_per_rollout"],
           note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of one rollout (bench.py --steps 1 --warmup 0), "
                "summed over every rollout_kernel dispatch (the reset-only launch included, it is tiny); KB units x1024, "
                "corrected by the factors measured on a known byte count in this library's access shape "
                "(profiles/r03_counter_calibration.json: FETCH_SIZE x2.0, WRITE_SIZE x1.0)")
json.dump(rec, open(f"gpurun_out/{tag}_hbm_traffic.json", "w"), indent=1)
json.dump(rec, open(f"gpurun_out/latest_{wl}_hbm_traffic.json", "w"), indent=1)  # what bench.py looks for under profiles/
print(rec)
PY
bash tools/pmc_sq.sh "$tag" "$@" | cut -c1-400
