cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
rm -rf gpurun_out/pc3 && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pc3 -o t -- python3 bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline --verify 0 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
rows=list(csv.DictReader(open(glob.glob('gpurun_out/pc3/**/*kernel_trace.csv',recursive=True)[0])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
sel=[r for r in rows if 'rollout_kernel' in r['Kernel_Name'] or 'control' in r['Kernel_Name']]
# last rollout: find last reset kernel (rollout_kernel<64, 1,...)
idx=[i for i,r in enumerate(sel) if 'rollout_kernel<' in r['Kernel_Name']][-1]
t0=int(sel[idx]['Start_Timestamp'])
for r in sel[idx:]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    print(r['Kernel_Name'][:24].replace('void sg::','').replace('sg::',''), 'q', r['Queue_Id'], round((s-t0)/1e6,3), round((e-t0)/1e6,3), 'dur', round((e-s)/1e6,3))
PY
rm -rf gpurun_out/pc3
