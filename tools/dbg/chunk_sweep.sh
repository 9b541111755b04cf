cd "$GRAFT_REPO_ROOT"
val() { grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,2))"; }
for rep in 1 2; do for c in 1024 256 384 512 640 768; do echo -n "rep $rep SG_CHUNK_STEPS=$c: "; SG_CHUNK_STEPS=$c python3 bench.py --no-cpu-baseline --verify 0 --steps 8 --warmup 2 2>/dev/null | val; done; done
