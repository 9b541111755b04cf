cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
b() { lab=$1; shift
  env "$@" timeout 300 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --verify 0 | python3 -c "import json,sys;l=json.loads(sys.stdin.read());r=l['roofline'];print('$lab', round(l['value']/1e9,2), round(l['ms_per_step'],2), r['launches_per_rollout'])"
}
b "default" X=1
b "chunk2048" SG_CHUNK_STEPS=2048
b "chunk512" SG_CHUNK_STEPS=512
b "chunk4096" SG_CHUNK_STEPS=4096
b "default again" X=1
