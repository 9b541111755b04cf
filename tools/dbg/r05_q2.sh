#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export SG_QUEUE_TIMEOUT_MS=8000
timeout 200 python tools/dbg/queue_timeline.py 2>&1 | grep -v "^t = " | tail -32
for g in 140 170; do for f in 64 96; do
SG_QUEUE_GROW=$g SG_QUEUE_FIRST=$f timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --verify 4 > gpurun_out/r05b_c3_g${g}_f$f.json 2>/dev/null
python -c "import json;l=json.load(open('gpurun_out/r05b_c3_g${g}_f$f.json'));print('grow $g first $f', round(l['value']/1e9,2), l['roofline']['schedule']['chunks'], l['verified']['equal'])"
done; done
bash tools/dbg/r05_gpu_tests.sh r05b
