#!/bin/bash
# interleaved A/B of environment variants on the c3 line: bash tools/dbg/r05_ab.sh "<envA>" "<envB>" ... (each a quoted list of VAR=VAL);
# BENCH_EXTRA: more bench.py arguments (e.g. "--verify 0" for variants that are measurement only)
cat > /tmp/ab_line.py <<'PY'
import json, sys
l = json.loads(sys.stdin.readlines()[-1])
v = l.get("verified") or {}
print('%-60s' % sys.argv[1], round(l['value'] / 1e9, 2), round(l['ms_per_step'], 2), round(l['roofline']['kernel_ms'], 2), l['roofline']['schedule'].get('chunks'), v.get('equal'))
PY
run() { ( for kv in $1; do export $kv; done; python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline $BENCH_EXTRA 2>/dev/null | python3 /tmp/ab_line.py "$1" ); }
for rep in $(seq 1 ${REPS:-3}); do for v in "$@"; do run "$v"; done; done
