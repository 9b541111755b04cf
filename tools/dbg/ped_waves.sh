#!/bin/bash
# general pedestrian variants: 2 wavefronts per SIMD with spills (product) against 1 wavefront per SIMD spill-free (ab/ped1.so)
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --no-cpu-baseline --verify 0 --steps 2 --warmup 1 "$@" 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'].get('kernel'))"; }
for lib in "" scenario_gym_amd/lib/ab/ped1.so; do
  export SGYM_LIB=$lib
  echo "== lib=${lib:-product}"
  echo -n "crowd kernel, c5 2000 steps: "; run --workload c5 --sim-steps 2000
  echo -n "general ped kernel, c5 2000 steps: "; SG_CROWD_KERNEL=0 run --workload c5 --sim-steps 2000
  echo -n "general ped kernel, 4096 x 64, 2000 steps: "; SG_CROWD_KERNEL=0 run --workload c5 --sim-steps 2000 --scenarios 4096 --entities 64
  echo -n "general ped kernel, 2048 x 128, 2000 steps: "; SG_CROWD_KERNEL=0 run --workload c5 --sim-steps 2000 --scenarios 2048 --entities 128
done
