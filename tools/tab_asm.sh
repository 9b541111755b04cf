#!/bin/bash
# Device assembly of one kernel of sgym_hip.hip: tools/tab_asm.sh [mangled-name prefix] -> /tmp/t/kernel.s (+ loop summary)
pat=${1:-_ZN2sg18rollout_kernel_tabILi64EE}
mkdir -p /tmp/t; cd "$(dirname "$0")/../scenario_gym_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-array-bounds -Wno-bitwise-instead-of-logical -Wno-unused-command-line-argument -mllvm --disable-promote-alloca-to-lds -S --offload-device-only -o /tmp/t/all.s sgym_hip.hip 2>&1 | grep -i " error"
S=$(grep -n "^$pat.*:" /tmp/t/all.s | head -1 | cut -d: -f1)
awk -v s=$S 'NR>=s' /tmp/t/all.s | awk '/s_endpgm/{print; exit} {print}' > /tmp/t/kernel.s
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
for a,b,k in sorted(back,key=lambda x:x[0]-x[1])[:8]:
    ins=[x for x in lines[a:b+1] if x.startswith('\t') and not x.strip().startswith(('.',';'))]
    c=lambda p: sum(1 for x in ins if re.match(r'\s+'+p,x))
    print(f"lines {a}-{b} {k}: {len(ins)} instr, valu {c('v_')} (mov_b64 {c('v_mov_b64')}, mov_b32 {c('v_mov_b32')}, cndmask {c('v_cndmask')}, readlane {c('v_readlane')}) salu {c('s_')} ds {c('ds_')}")
PY
