cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for R in 512 1024; do python3 bench.py --workload c3s --scenarios $R --steps 10 --warmup 2 --no-cpu-baseline | python3 -c "import json,sys;l=json.loads(sys.stdin.read());print('c3s $R', l['value'], l['ms_per_step'], l['verified']['equal'])"; done
python3 bench.py --workload c2s --scenarios 1024 --entities 64 --steps 10 --warmup 2 --no-cpu-baseline | python3 -c "import json,sys;l=json.loads(sys.stdin.read());print('replay 1024x64 sliced', l['value'], l['ms_per_step'], l['verified']['equal'])"
python3 bench.py --workload c2 --scenarios 1024 --entities 64 --steps 10 --warmup 2 --no-cpu-baseline | python3 -c "import json,sys;l=json.loads(sys.stdin.read());print('replay 1024x64 stepwise', l['value'], l['ms_per_step'], l['verified']['equal'])"
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sliced or prepass" 2>&1 | tail -3
