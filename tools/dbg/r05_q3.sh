#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export SG_QUEUE_TIMEOUT_MS=8000
timeout 200 python tools/dbg/queue_timeline.py 2>&1 | grep -v "^t = " | tail -26
for i in 1 2 3; do
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --verify 4 > gpurun_out/r05c_c3_$i.json 2>/dev/null
python -c "import json;l=json.load(open('gpurun_out/r05c_c3_$i.json'));print('c3 run $i', round(l['value']/1e9,2), l['roofline']['schedule']['chunks'], l['verified']['equal'])"
done
for g in 170 200; do
SG_QUEUE_GROW=$g timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --verify 4 > gpurun_out/r05c_c3_g${g}.json 2>/dev/null
python -c "import json;l=json.load(open('gpurun_out/r05c_c3_g${g}.json'));print('grow $g', round(l['value']/1e9,2), l['roofline']['schedule']['chunks'], l['verified']['equal'])"
done
bash tools/dbg/r05_gpu_tests.sh r05c
