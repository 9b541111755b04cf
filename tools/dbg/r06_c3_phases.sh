#!/bin/bash
# GPU box: where the cycles of a step of the table kernels go (tools/ab_build.sh phase -DSG_PHASE_TIMERS first): s_memtime between
# the PH(i) marks of rollout_body_l, summed over wavefronts.  512 scenarios = one wavefront per SIMD (the chain of a lone
# wavefront, phase by phase), 4096 = three per SIMD (c3), c2 = 64 lone wavefronts of 16-lane tiles.
# -> gpurun_out/r06_c3_phase_cycles.txt
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
out=gpurun_out/r06_c3_phase_cycles.txt
cat > $out <<'TXT'
phases of a step of the table kernels (experiment build -DSG_PHASE_TIMERS: a mark costs an s_memtime + a scalar add; shares of the wavefronts' resident cycles)
 [5] loop top, run test, appearance, table row / replay interpolation ... up to the vote; and behind the collision pass: events, terminal conditions, statistics, row stores
 [7] run vote   [1] move: poses, velocities, distance, ego metrics
 [8] box centre / sin cos in fp32, cell index   [11] sync before publishing   [2] publishing the tile's terms in LDS, stripe atomics   [12] range vote
 [13] sync behind the atomics   [9] stripe-mask reads + circle test per candidate   [10] the all-pairs walk (tiles of <= 16 lanes)
 [3] fp32 SAT filter   [15] fuzzy vote   [4] exact fp64 SAT, owner mapping
TXT
for args in "--workload c3 --scenarios 512" "--workload c3" "--workload c2"; do
  echo "== $args" >> $out
  SGYM_LIB=scenario_gym_amd/lib/ab/phase.so python3 bench.py --no-cpu-baseline --no-configs $args --steps 1 --warmup 0 2>&1 >/dev/null | grep "phase cycles" | tail -1 >> $out
done
cat $out
