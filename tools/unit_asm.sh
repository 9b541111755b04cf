#!/bin/bash
# Device assembly of one kernel of one unit of libsgym_hip.so:
#   tools/unit_asm.sh k_crowd '_ZN2sg20rollout_kernel_crowdILi4EE' [extra flags]  ->  /tmp/t/<unit>.s, /tmp/t/kernel.s + a loop summary
unit=${1:-k_crowd}; pat=${2:-_ZN2sg20rollout_kernel_crowdILi4EE}; shift 2
mkdir -p /tmp/t; cd "$(dirname "$0")/../scenario_gym_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-array-bounds -Wno-bitwise-instead-of-logical -Wno-unused-command-line-argument -mllvm --disable-promote-alloca-to-lds "$@" -S --offload-device-only -o /tmp/t/$unit.s $unit.hip 2>&1 | grep -i " error"
S=$(grep -n "^$pat.*:" /tmp/t/$unit.s | head -1 | cut -d: -f1)
awk -v s=$S 'NR>=s' /tmp/t/$unit.s | awk '/s_endpgm/{print; exit} {print}' > /tmp/t/kernel.s
python3 - <<'PY'
import re
lines=open('/tmp/t/kernel.s').read().split('\n')
lab={}
for i,l in enumerate(lines):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: lab[m.group(1)]=i
back=[]
for i,l in enumerate(lines):
    m=re.search(r'\s(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)',l)
    if m and m.group(2) in lab and lab[m.group(2)]<i:
        back.append((lab[m.group(2)],i,m.group(1)))
tot=[x for x in lines if x.startswith('\t') and not x.strip().startswith(('.',';'))]
print(f"kernel: {len(tot)} instr, scratch ops {sum(1 for x in tot if 'scratch_' in x)}")
for a,b,k in sorted(back,key=lambda x:x[0]-x[1])[:10]:
    ins=[x for x in lines[a:b+1] if x.startswith('\t') and not x.strip().startswith(('.',';'))]
    c=lambda p: sum(1 for x in ins if re.match(r'\s+'+p,x))
    print(f"lines {a}-{b} {k}: {len(ins)} instr, valu {c('v_')} (mov_b64 {c('v_mov_b64')}, mov_b32 {c('v_mov_b32')}, cndmask {c('v_cndmask')}) salu {c('s_')} ds {c('ds_')} scratch {c('scratch_')} barrier {c('s_barrier')}")
PY
