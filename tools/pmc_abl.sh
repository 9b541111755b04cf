cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in libsgym_hip libsgym_abl_NO_NARROW libsgym_abl_NO_COLL libsgym_abl_NO_COLLNO_SINCOS; do
  export SGYM_LIB=$PWD/scenario_gym_amd/lib/$lib.so
  timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/abl_$lib -o pmc -- python3 bench.py --steps 1 --warmup 0 --sim-steps 2000 --no-cpu-baseline > gpurun_out/abl_$lib.log 2>&1
  f=$(find gpurun_out/abl_$lib -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" $lib <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(float)
for r in rows:
    if 'rollout_kernel' in r['Kernel_Name']:
        agg[(r['Dispatch_Id'],r['Counter_Name'])]+=float(r['Counter_Value'])
d=sorted(set(k[0] for k in agg),key=int)[-1]
ws=4096*2000
print(sys.argv[2],{k[1]:round(v/ws,1) for k,v in agg.items() if k[0]==d})
PY
done
