#!/bin/bash
# c5 (1024 x 256 crowd) under several builds / switches: tools/dbg/r06_c5_libs.sh "<lib|-> <ENV=..|->" ...   (lib = name under scenario_gym_amd/lib/ab/)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
i=0
for spec in "$@"; do
  set -- $spec; lib=$1; envs=$2; i=$((i+1))
  L=""; [ "$lib" != "-" ] && L="SGYM_LIB=scenario_gym_amd/lib/ab/$lib.so"
  E=""; [ "$envs" != "-" ] && E=$(echo $envs | tr ',' ' ')
  env $L $E timeout 900 python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --verify 2 > gpurun_out/r06_c5_libs_$i.json 2> gpurun_out/r06_c5_libs_$i.err
  python -c "import json;l=json.load(open('gpurun_out/r06_c5_libs_$i.json'));print('$lib $envs:', round(l['value']/1e9,3), 'G', round(l['ms_per_step'],1), 'ms', l['verified']['equal'])" || tail -5 gpurun_out/r06_c5_libs_$i.err
done
