cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline | python3 -c "import json,sys;l=json.loads(sys.stdin.read());print('c3 planar', l['value'], l['ms_per_step'], l['roofline']['kernel_ms'], l['verified']['equal'])"
SG_PLANAR=0 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline | python3 -c "import json,sys;l=json.loads(sys.stdin.read());print('c3 general', l['value'], l['ms_per_step'], l['roofline']['kernel_ms'], l['verified']['equal'])"
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
SG_PLANAR=0 timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
