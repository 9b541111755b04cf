#!/bin/bash
# round 5, first contact of the persistent table launch with the GPU: the new tests, then c3 with and without the queue
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export SG_QUEUE_TIMEOUT_MS=5000
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "queue_launch or late_spawns or launch_stats or prepass" > gpurun_out/r05a_queue_tests.txt 2>&1
tail -5 gpurun_out/r05a_queue_tests.txt
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r05a_c3_queue.json 2> gpurun_out/r05a_c3_queue.err
tail -c 600 gpurun_out/r05a_c3_queue.err; python -c "import json;l=json.load(open('gpurun_out/r05a_c3_queue.json'));print('queue', l['value']/1e9, l['roofline']['schedule'], l['verified'])"
SG_QUEUE=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r05a_c3_chunks.json 2> gpurun_out/r05a_c3_chunks.err
python -c "import json;l=json.load(open('gpurun_out/r05a_c3_chunks.json'));print('chunks', l['value']/1e9, l['roofline']['schedule']['per_rank'], l['verified'])"
