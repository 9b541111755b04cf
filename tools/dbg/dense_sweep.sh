cd "$GRAFT_REPO_ROOT"
val() { grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,3), round(d['roofline']['kernel_ms'],1))"; }
echo -n "product: "; python3 bench.py --workload c5 --no-cpu-baseline --verify 0 --steps 2 --warmup 1 2>/dev/null | val
for n in g h i j k l; do echo -n "dn_$n: "; SGYM_LIB=scenario_gym_amd/lib/ab/dn_$n.so python3 bench.py --workload c5 --no-cpu-baseline --verify 0 --steps 2 --warmup 1 2>/dev/null | val; done
echo -n "product again: "; python3 bench.py --workload c5 --no-cpu-baseline --verify 0 --steps 2 --warmup 1 2>/dev/null | val
