cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
b() { lab=$1; R=$2; shift; shift
  env "$@" timeout 300 python3 bench.py --scenarios $R --steps 3 --warmup 1 --no-cpu-baseline --verify 4 | python3 -c "import json,sys;l=json.loads(sys.stdin.read());r=l['roofline'];print('$lab R=$R', round(l['value']/1e9,2), round(l['ms_per_step'],2), l['verified']['equal'], r['launches_per_rollout'], round(r['kernel_ms'],3), round(r['kernel_ms_gross'],3))"
}
b "default" 4096 X=1
b "Q4" 4096 GPU_MAX_HW_QUEUES=4
b "Q3" 4096 GPU_MAX_HW_QUEUES=3
b "Q2" 4096 GPU_MAX_HW_QUEUES=2
b "Q1" 4096 GPU_MAX_HW_QUEUES=1
b "split1" 4096 SG_TAB_SPLIT=1
timeout 900 python3 -m pytest tests -x -q -m gpu --timeout 300 2>&1 | tail -3
