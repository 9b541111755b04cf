cd "$GRAFT_REPO_ROOT"
val() { grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['stage_seconds'])"; }
for cfg in "512 32" "512 16" "512 64" "256 32" "384 24" "1024 32" "512 32"; do set -- $cfg; echo -n "chunk $1 sub $2: "; SG_E2E_SUB=$2 python3 bench.py --workload e2e --chunk $1 --no-cpu-baseline 2>/dev/null | val; done
