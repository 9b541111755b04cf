#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "baseline (walk off):"; SG_CROWD_WALK=0 python3 tools/dbg/walk_pmc.py 2>&1 | tail -1 | grep -o '"kernel_ms": [0-9.]*'
for cfg in "3 200 64" "3 400 64" "3 100 64" "3 200 40" "3 200 28" "2 200 64" "1 200 64"; do
  set -- $cfg
  echo -n "walk=$1 chunk=$2 walk1_max=$3: "; SG_CROWD_WALK=$1 SG_CROWD_CHUNK=$2 SG_WALK1_MAX=$3 python3 tools/dbg/walk_pmc.py 2>&1 | tail -1 | grep -o '"walk1": [0-9]*\|"walk2": [0-9]*\|"kernel_ms": [0-9.]*' | tr "\n" " "; echo
done
