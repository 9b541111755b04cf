cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
df -h /dev/shm /tmp | tail -2
timeout 900 python3 bench.py --workload e2e --files 4096 > gpurun_out/r03_e2e_bench.json 2> gpurun_out/r03_e2e.err; tail -5 gpurun_out/r03_e2e.err; python3 -c "
import json;l=json.load(open('gpurun_out/r03_e2e_bench.json'));print(l['value'], l['x_realtime'], l['stage_seconds'], l['ms_per_step'], l.get('cpu_baseline'), l['sample_metrics'])"
timeout 600 python3 -m pytest tests/test_ingest_json.py -x -q -m gpu 2>&1 | tail -3
