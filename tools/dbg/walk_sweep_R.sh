#!/bin/bash
# walker dispatch against the fused crowd kernel at larger batches (more than one walker wavefront per SIMD)
cd "$GRAFT_REPO_ROOT"
for R in 2048 4096; do
  for w in 0 3 1; do
    echo -n "R=$R walk=$w: "; R=$R T=4000 SG_CROWD_WALK=$w python3 tools/dbg/walk_pmc.py 2>&1 | tail -1 | grep -o '"walk1": [0-9]*\|"walk2": [0-9]*\|"kernel_ms": [0-9.]*' | tr "\n" " "; echo
  done
done
