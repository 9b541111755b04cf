cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/wide_prof -o w -- python3 tools/dbg/wide_time.py > /dev/null 2>&1
g=$(find gpurun_out/wide_prof -name "*kernel_trace.csv" | head -1)
python3 - $g <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"]
    if "wide_" in n:
        key=(n.split("(")[0].replace("sg::",""), r.get("Grid_Size_X","?"), r.get("Grid_Size_Y","?"))
        agg[key].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(agg.items()): print(k, len(v), "avg us", round(sum(v)/len(v),1), "max", round(max(v),1))
PY
rm -rf gpurun_out/wide_prof
