#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
tag=${1:-r04w}
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM --output-format csv -d gpurun_out/${tag}_a -o p -- python3 tools/dbg/walk_pmc.py > gpurun_out/${tag}_a.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_IFETCH SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/${tag}_b -o p -- python3 tools/dbg/walk_pmc.py > gpurun_out/${tag}_b.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
st = json.loads([l for l in open(f"gpurun_out/{tag}_a.log") if l.startswith("STATS")][-1][6:])
agg = collections.defaultdict(float)
for f in glob.glob(f"gpurun_out/{tag}_a/**/*counter_collection.csv", recursive=True)[:1] + glob.glob(f"gpurun_out/{tag}_b/**/*counter_collection.csv", recursive=True)[:1]:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "walk1" if ("walk_kernel<1>" in n or "walk4_kernel" in n) else "walk2" if "walk_kernel<2>" in n else "full" if "rollout_kernel_crowd" in n else None
        if k: agg[(k, r["Counter_Name"])] += float(r["Counter_Value"])
ws = {"walk1": st["walk1"] * st["chunk"] * 1, "walk2": st["walk2"] * st["chunk"] * 2, "full": st["full"] * st["chunk"] * 4}
out = {k: {c[1]: round(v / max(ws[k], 1), 1) for c, v in agg.items() if c[0] == k} for k in ws}
print(json.dumps(dict(stats=st, wave_steps=ws, per_wave_step=out), indent=1))
json.dump(dict(stats=st, wave_steps=ws, per_wave_step=out), open(f"gpurun_out/{tag}_pmc.json", "w"), indent=1)
PY
rm -rf gpurun_out/${tag}_a gpurun_out/${tag}_b
