#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export SG_QUEUE_TIMEOUT_MS=8000
for i in 1 2; do
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --verify 16 > gpurun_out/r05d_c3_$i.json 2>/dev/null
python -c "import json;l=json.load(open('gpurun_out/r05d_c3_$i.json'));print('c3 run $i', round(l['value']/1e9,2), l['roofline']['kernel_ms'], l['verified']['equal'])"
done
timeout 300 python bench.py --workload c2 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r05d_c2.json 2>/dev/null
python -c "import json;l=json.load(open('gpurun_out/r05d_c2.json'));print('c2', round(l['value']/1e9,3), l['verified']['equal'])"
timeout 300 python bench.py --workload c3s --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r05d_c3s.json 2>/dev/null
python -c "import json;l=json.load(open('gpurun_out/r05d_c3s.json'));print('c3s', round(l['value']/1e9,3), l['verified']['equal'])"
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "not walker_variant" 2>&1 | tail -4
