#!/bin/bash
# chunk lengths of the persistent launch (measurement): bash tools/dbg/r05_cap.sh
mkdir -p gpurun_out
run() { python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.readlines()[-1]); print('$1', round(l['value']/1e9,2), l['ms_per_step'], l['roofline']['schedule'].get('chunks'), l['verified']['equal'])"; }
run default
SG_QUEUE_DECAY=0 run decay0
SG_QUEUE_DECAY=120 run decay120
SG_QUEUE_DECAY=200 run decay200
SG_QUEUE_CAP=640 run cap640
SG_QUEUE_CAP=768 run cap768
SG_QUEUE_CAP=448 run cap448
SG_QUEUE_GROW=125 run grow125
run default
python3 bench.py --workload c3rss --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.readlines()[-1]); print('c3rss', round(l['value']/1e9,2), l['ms_per_step'], l['roofline']['schedule'].get('chunks'), l['verified']['equal'])"
for R in 512 1024 2048; do python3 bench.py --scenarios $R --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.readlines()[-1]); print('shard $R', round(l['value']/1e9,2), l['ms_per_step'], l['roofline']['schedule'].get('chunks'), l['verified']['equal'])"; done
