#!/bin/bash
# Usage (GPU box): bash tools/hbm_calib.sh <tag>  -> gpurun_out/<tag>_counter_calibration.json (copy to profiles/)
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
exe=scenario_gym_amd/lib/hbm_calib
[ -x $exe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $exe tools/hbm_calib.hip
./$exe 2 > gpurun_out/${tag}_calib_plain.json
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/${tag}_calib_$c -o p -- ./$exe 2 > gpurun_out/${tag}_calib_$c.log 2>&1
done
python3 - "$tag" <<'PY'
import csv, glob, json, sys
tag = sys.argv[1]
plain = json.loads(open(f"gpurun_out/{tag}_calib_plain.json").read().strip().splitlines()[-1])
n = plain["bytes_per_kernel"]
rec = {"bytes_per_kernel": n, "timing": plain, "counters_kb": {}, "access_shape": "512-byte rows, 8 B per lane, one row per wavefront instruction"}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/{tag}_calib_{c}/**/*counter_collection.csv", recursive=True)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == c and "calib_" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0]
            rec["counters_kb"].setdefault(k, {})[c] = float(r["Counter_Value"])
ck = rec["counters_kb"]
rec["fetch_bytes_per_counted_kb"] = n / ck["calib_read"]["FETCH_SIZE"]          # true bytes per reported KB
rec["write_bytes_per_counted_kb"] = n / ck["calib_write"]["WRITE_SIZE"]
rec["fetch_factor"] = rec["fetch_bytes_per_counted_kb"] / 1024.0                  # multiply FETCH_SIZE x 1024 by this
rec["write_factor"] = rec["write_bytes_per_counted_kb"] / 1024.0
rec["copy_check"] = {"fetch": ck["calib_copy"]["FETCH_SIZE"] * 1024 * rec["fetch_factor"] / n,
                     "write": ck["calib_copy"]["WRITE_SIZE"] * 1024 * rec["write_factor"] / n}
json.dump(rec, open(f"gpurun_out/{tag}_counter_calibration.json", "w"), indent=1)
print(json.dumps(rec))
PY
