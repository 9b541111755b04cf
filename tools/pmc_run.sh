cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 120 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_$n -o pmc -- python3 bench.py --steps 1 --warmup 0 --sim-steps 2000 --no-cpu-baseline > gpurun_out/pmc_$n.log 2>&1
  f=$(find gpurun_out/pmc_$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(float)
for r in rows:
    if 'rollout_kernel' in r['Kernel_Name'] and int(r.get('Grid_Size',0) or 0)>0:
        agg[(r['Dispatch_Id'],r['Counter_Name'])]+=float(r['Counter_Value'])
# print last dispatch (the big one)
disp=sorted(set(k[0] for k in agg),key=int)
for d in disp[-1:]:
    print('dispatch',d,{k[1]:v for k,v in agg.items() if k[0]==d})
PY
done
