cd "$GRAFT_REPO_ROOT"
val() { grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,3))"; }
echo -n "c5: "; python3 bench.py --workload c5 --no-cpu-baseline --verify 2 --steps 2 --warmup 1 2>/dev/null | val
echo -n "c5mix: "; python3 bench.py --workload c5mix --no-cpu-baseline --verify 2 --steps 1 --warmup 1 2>/dev/null | val
echo -n "c5 noise: "; python3 bench.py --workload c5 --ped-noise device --no-cpu-baseline --verify 0 --steps 2 --warmup 1 2>/dev/null | val
echo -n "crowd 4096 x 64 x 2000: "; python3 bench.py --workload c5 --scenarios 4096 --entities 64 --sim-steps 2000 --no-cpu-baseline --verify 0 --steps 2 --warmup 1 2>/dev/null | val
echo -n "crowd 2048 x 128 x 2000: "; python3 bench.py --workload c5 --scenarios 2048 --entities 128 --sim-steps 2000 --no-cpu-baseline --verify 0 --steps 2 --warmup 1 2>/dev/null | val
echo -n "general ped kernel c5 2000: "; SG_CROWD_KERNEL=0 python3 bench.py --workload c5 --sim-steps 2000 --no-cpu-baseline --verify 0 --steps 2 --warmup 1 2>/dev/null | val
