cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sliced or prepass" 2>&1 | tail -4
python3 bench.py --workload c3s --steps 10 --warmup 2 --no-cpu-baseline | python3 -c "import json,sys;l=json.loads(sys.stdin.read());print('fast', l['value'], l['ms_per_step'], l['verified']['equal'])"
SG_CTL_FAST=0 python3 bench.py --workload c3s --steps 10 --warmup 2 --no-cpu-baseline --verify 0 | python3 -c "import json,sys;l=json.loads(sys.stdin.read());print('general', l['value'], l['ms_per_step'])"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03c_c3s_trace -o t -- python3 bench.py --workload c3s --steps 3 --warmup 1 --no-cpu-baseline --verify 0 > /dev/null 2>&1
f=$(find gpurun_out/r03c_c3s_trace -name "*kernel_stats.csv" | head -1); cut -c1-160 $f | head -5
