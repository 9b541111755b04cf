#!/bin/bash
# GPU box: the staggered crowd schedule (launch_crowd_staggered) against one launch over the batch, on c5 and its neighbours.
# Needs tools/experiments/r06_crowd_stagger.patch applied and the library rebuilt (the product has no such schedule).
# Usage: bash tools/dbg/r06_stagger_sweep.sh  -> gpurun_out/r06_stagger_sweep.txt
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
out=gpurun_out/r06_stagger_sweep.txt; : > $out
run() { # label, env..., -- bench args
    local label=$1; shift
    local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    env "${envs[@]}" python3 bench.py --no-cpu-baseline --no-configs --steps 3 --warmup 1 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']
print('%-44s %8.3f G  %8.2f ms  launches %s  verified %s' % ('$label', d['value']/1e9, d['ms_per_step'], r.get('launches_per_rollout'), (d.get('verified') or {}).get('equal')))" >> $out
}
for rep in 1 2; do
run "c5 one launch"            SG_CROWD_STAGGER=0 -- --workload c5
run "c5 staggered 50%"         SG_CROWD_STAGGER=1 -- --workload c5
run "c5 staggered 35%"         SG_CROWD_STAGGER_PCT=35 -- --workload c5
run "c5 staggered 60%"         SG_CROWD_STAGGER_PCT=60 -- --workload c5
run "c5 staggered 70%"         SG_CROWD_STAGGER_PCT=70 -- --workload c5
done
run "c5roads one launch"       SG_CROWD_STAGGER=0 -- --workload c5roads
run "c5roads staggered"        SG_CROWD_STAGGER=1 -- --workload c5roads
run "c5 device noise one launch" SG_CROWD_STAGGER=0 -- --workload c5 --ped-noise device
run "c5 device noise staggered"  SG_CROWD_STAGGER=1 -- --workload c5 --ped-noise device
run "2048x128 one launch"      SG_CROWD_STAGGER=0 -- --workload c5 --scenarios 2048 --entities 128
run "2048x128 staggered"       SG_CROWD_STAGGER=1 -- --workload c5 --scenarios 2048 --entities 128
run "4096x64 one launch"       SG_CROWD_STAGGER=0 -- --workload c5 --scenarios 4096 --entities 64
run "4096x64 staggered"        SG_CROWD_STAGGER=1 -- --workload c5 --scenarios 4096 --entities 64
run "2048x256 one launch"      SG_CROWD_STAGGER=0 -- --workload c5 --scenarios 2048
run "2048x256 staggered"       SG_CROWD_STAGGER=1 -- --workload c5 --scenarios 2048
run "1536x256 one launch"      SG_CROWD_STAGGER=0 -- --workload c5 --scenarios 1536
run "1536x256 staggered"       SG_CROWD_STAGGER=1 -- --workload c5 --scenarios 1536
cat $out
