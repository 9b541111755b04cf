cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 900 python3 bench.py --workload e2e --files 4096 > gpurun_out/r03_e2e_bench.json 2> gpurun_out/r03_e2e.err; tail -3 gpurun_out/r03_e2e.err; python3 -c "
import json;l=json.load(open('gpurun_out/r03_e2e_bench.json'));print(l['value'], l['x_realtime'], l['stage_seconds'], l['ms_per_step'])"
timeout 900 python3 bench.py --workload c5mix --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03_c5mix_bench.json 2> gpurun_out/r03_c5mix.err; tail -3 gpurun_out/r03_c5mix.err; python3 -c "
import json;l=json.load(open('gpurun_out/r03_c5mix_bench.json'));print('c5mix', l['value'], l['ms_per_step'], l['verified'])"
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sliced" 2>&1 | tail -3
