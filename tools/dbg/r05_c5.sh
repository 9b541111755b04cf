#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for m in ${MASKS:-0 3 4 6}; do for ch in ${CHUNKS:-200}; do
SG_CROWD_WALK=$m SG_CROWD_CHUNK=$ch timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --verify 2 > gpurun_out/r05_c5_m${m}_c$ch.json 2> gpurun_out/r05_c5_m${m}_c$ch.err
python -c "import json;l=json.load(open('gpurun_out/r05_c5_m${m}_c$ch.json'));print('walk mask $m chunk $ch:', round(l['value']/1e9,3), 'G', round(l['ms_per_step'],1), 'ms', l['verified']['equal'])" || tail -3 gpurun_out/r05_c5_m${m}_c$ch.err
done; done
