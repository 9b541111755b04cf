cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for i in 1 2; do python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline | python3 -c "import json,sys;l=json.loads(sys.stdin.read());print('c3 planar', l['value'], l['ms_per_step'], l['roofline']['kernel_ms'], l['verified']['equal'])"; done
SG_PLANAR=0 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline | python3 -c "import json,sys;l=json.loads(sys.stdin.read());print('c3 general', l['value'], l['ms_per_step'], l['roofline']['kernel_ms'], l['verified']['equal'])"
python3 bench.py --workload c3rss --steps 3 --warmup 1 --no-cpu-baseline | python3 -c "import json,sys;l=json.loads(sys.stdin.read());print('c3rss', l['value'], l['ms_per_step'], l['verified']['equal'])"
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
