cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sliced" 2>&1 | tail -15
for R in 512 1024; do
  timeout 300 python3 bench.py --scenarios $R --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r03b_shard_${R}_bench.json 2> gpurun_out/r03b_shard_${R}.err
  python3 -c "import json;l=json.load(open('gpurun_out/r03b_shard_${R}_bench.json'));print($R, l['value'], l['ms_per_step'], l['roofline']['kernel_ms'], l['roofline']['launches_per_rollout'], l['verified']['equal'] if l.get('verified') else None)"
  tail -3 gpurun_out/r03b_shard_${R}.err
done
