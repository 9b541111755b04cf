#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for lib in $LIBS; do for m in ${MASKS:-4}; do for ch in ${CHUNKS:-200}; do
SGYM_LIB=scenario_gym_amd/lib/ab/$lib.so SG_CROWD_WALK=$m SG_CROWD_CHUNK=$ch timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --verify 2 > gpurun_out/r05_c5_ab.json 2> gpurun_out/r05_c5_ab.err
python -c "import json;l=json.load(open('gpurun_out/r05_c5_ab.json'));print('$lib walk mask $m chunk $ch:', round(l['value']/1e9,3), 'G', round(l['ms_per_step'],1), 'ms', l['verified']['equal'])" || tail -3 gpurun_out/r05_c5_ab.err
done; done; done
