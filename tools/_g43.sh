cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
b() { lab=$1; R=$2; shift; shift
  env "$@" timeout 300 python3 bench.py --scenarios $R --steps 3 --warmup 1 --no-cpu-baseline --verify 4 | python3 -c "import json,sys;l=json.loads(sys.stdin.read());r=l['roofline'];print('$lab R=$R', round(l['value']/1e9,2), round(l['ms_per_step'],2), l['verified']['equal'], r['launches_per_rollout'], round(r['kernel_ms'],3), round(r['kernel_ms_gross'],3))"
}
for H in 2 3 4; do b "Q8 H=$H" 4096 SG_TAB_SPLIT=$H GPU_MAX_HW_QUEUES=8; done
for H in 3 4; do b "Q8 H=$H" 8192 SG_TAB_SPLIT=$H GPU_MAX_HW_QUEUES=8; done
b "Q2 H=2" 4096 SG_TAB_SPLIT=2 GPU_MAX_HW_QUEUES=2
