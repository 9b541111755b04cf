cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
rm -rf gpurun_out/up_trace
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/up_trace -o t -- python3 tools/upload_time.py > gpurun_out/up_trace.log 2>&1
f=$(find gpurun_out/up_trace -name "*kernel_stats.csv" | head -1)
head -12 "$f" | cut -c1-160
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/up_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[:60]:
    print(f'{(int(r["Start_Timestamp"])-t0)/1e6:10.3f} ms  {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:9.1f} us  {r["Kernel_Name"][:70]}')
PY
