#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export SG_QUEUE_TIMEOUT_MS=8000
for ps in 0 1500 3000 5000; do for i in 1 2; do
SG_QUEUE_PARK_STEPS=$ps timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --verify 4 > gpurun_out/r05f_c3.json 2>/dev/null
python -c "import json;l=json.load(open('gpurun_out/r05f_c3.json'));print('park $ps run $i', round(l['value']/1e9,2), round(l['roofline']['kernel_ms'],2), l['verified']['equal'])"
done; done
SG_QUEUE_PARK_STEPS=3000 timeout 200 python tools/dbg/queue_timeline.py 2>&1 | grep -v "^t = " | tail -24
