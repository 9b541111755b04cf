cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -12
timeout 900 python3 bench.py --workload c5mix --steps 2 --warmup 1 --no-cpu-baseline | python3 -c "import json,sys;l=json.loads(sys.stdin.read());print('c5mix riders', l['value'], l['ms_per_step'], l['verified']['equal'], l['verified']['mismatches'])"
python3 bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline | python3 -c "import json,sys;l=json.loads(sys.stdin.read());print('c5', l['value'], l['ms_per_step'], l['verified']['equal'])"
