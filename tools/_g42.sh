cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
b() { lab=$1; R=$2; shift; shift
  env "$@" timeout 300 python3 bench.py --scenarios $R --steps 3 --warmup 1 --no-cpu-baseline --verify 4 | python3 -c "import json,sys;l=json.loads(sys.stdin.read());r=l['roofline'];print('$lab R=$R', round(l['value']/1e9,2), round(l['ms_per_step'],2), l['verified']['equal'], r['launches_per_rollout'], round(r['kernel_ms'],3), round(r['kernel_ms_gross'],3))"
}
for H in 2 3 4; do b "H=$H" 4096 SG_TAB_SPLIT=$H; done
for R in 512 1024 2048 8192; do for H in 1 2 4; do b "H=$H" $R SG_TAB_SPLIT=$H SG_TAB_SPLIT_MIN=2 SG_SLICE=0; done; done
