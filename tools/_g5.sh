cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 600 python3 -m pytest tests -x -q -m gpu --timeout 120 2>&1 | tail -5
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline | python3 -c "import json,sys;l=json.loads(sys.stdin.read());print('c3', l['value'], l['ms_per_step'], l['verified']['equal'])"
python3 bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline | python3 -c "import json,sys;l=json.loads(sys.stdin.read());print('c5', l['value'], l['ms_per_step'], l['verified']['equal'])"
