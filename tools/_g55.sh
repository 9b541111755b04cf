cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
b() { lab=$1; shift
  env "$@" timeout 300 python3 bench.py --steps 8 --warmup 1 --no-cpu-baseline --verify 0 | python3 -c "import json,sys;l=json.loads(sys.stdin.read());r=l['roofline'];print('$lab', round(l['value']/1e9,2), round(l['ms_per_step'],2), r['launches_per_rollout'])"
}
b "c1024" X=1
b "c768" SG_CHUNK_STEPS=768
b "c512" SG_CHUNK_STEPS=512
b "c384" SG_CHUNK_STEPS=384
b "c256" SG_CHUNK_STEPS=256
b "c512 slice32" SG_CHUNK_STEPS=512 SG_CTL_SLICE=32
b "c512 rss" SG_CHUNK_STEPS=512
