#!/bin/bash
# crowds on road networks: the new parity test, then c5 / c5roads (crowd kernel) and c5roads through the general kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "road_networks or long_crowd" 2>&1 | tail -6
for spec in "c5 -" "c5roads -" "c5roads SG_CROWD_ROADS=0"; do
  set -- $spec; w=$1; e=$2; E=""; [ "$e" != "-" ] && E=$e
  env $E timeout 900 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --verify 4 > gpurun_out/r06_roads_$w.json 2> gpurun_out/r06_roads_$w.err
  python -c "import json;l=json.load(open('gpurun_out/r06_roads_$w.json'));print('$w $e:', round(l['value']/1e9,3), 'G', round(l['ms_per_step'],1), 'ms', l['verified']['equal'], l['roofline']['kernel'])" || tail -5 gpurun_out/r06_roads_$w.err
done
