#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for wl in c3 c3rss c3s c2 c2s c5mix; do
timeout 600 python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r05g_$wl.json 2>gpurun_out/r05g_$wl.err
python -c "import json;l=json.load(open('gpurun_out/r05g_$wl.json'));print('$wl', round(l['value']/1e9,3), 'G', round(l['ms_per_step'],2), 'ms', l['roofline']['kernel'], (l['roofline'].get('schedule') or {}).get('per_rank'), l['verified']['equal'])" || tail -3 gpurun_out/r05g_$wl.err
done
