#!/bin/bash
# the c3 line without a process group and with RCCL initialised first (one rank), alternating, SG_CTL_PRIO on / off
cd "$GRAFT_REPO_ROOT"
val() { grep '^{' | tail -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['value']/1e9,1))"; }
for prio in 1 0; do for i in 1 2 3; do
  a=$(SG_CTL_PRIO=$prio python3 bench.py --steps 8 --warmup 2 --verify 0 --no-cpu-baseline 2>/dev/null | val)
  b=$(SG_CTL_PRIO=$prio SGYM_FORCE_DIST=1 MASTER_PORT=29533 python3 bench.py --steps 8 --warmup 2 --verify 0 --no-cpu-baseline 2>/dev/null | val)
  echo "prio=$prio plain $a rccl $b"
done; done
