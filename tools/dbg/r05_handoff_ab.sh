#!/bin/bash
# c3 with the three hand-off forms of the persistent launch (fence / write-through / none) + the queue tests with write-through
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export SG_QUEUE_TIMEOUT_MS=5000
for m in 1 0 2; do
SG_QUEUE_HANDOFF=$m timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r05a_c3_handoff$m.json 2> gpurun_out/r05a_c3_handoff$m.err
python -c "import json;l=json.load(open('gpurun_out/r05a_c3_handoff$m.json'));print('handoff $m', l['value']/1e9, l['verified']['equal'], l['verified']['mismatches'])"
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "queue_launch or late_spawns or launch_stats or prepass" > gpurun_out/r05a_queue_tests.txt 2>&1
tail -3 gpurun_out/r05a_queue_tests.txt
