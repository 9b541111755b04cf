#!/bin/bash
# c3 (4096 x 64, PID egos) under several builds, interleaved: tools/dbg/r06_c3_libs.sh <lib|-> ...   (lib = name under scenario_gym_amd/lib/ab/)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
for rep in 1 2 3; do
for lib in "$@"; do
  L=""; [ "$lib" != "-" ] && L="SGYM_LIB=scenario_gym_amd/lib/ab/$lib.so"
  env $L python bench.py --no-cpu-baseline --no-configs --verify 4 --steps 12 --warmup 3 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('$lib', round(l['value']/1e9,2), 'G', round(l['roofline']['kernel_ms'],2), 'ms', l['verified']['equal'])"
done
done
