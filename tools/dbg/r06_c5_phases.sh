#!/bin/bash
# GPU box: where the cycles of a c5 step go (tools/ab_build.sh phase -DSG_PHASE_TIMERS first): s_memtime between the PH(i)
# marks of rollout_kernel_crowd<4>, summed over wavefronts, one 10,000-step rollout; the same for c5roads.
# -> gpurun_out/r06_c5_phase_cycles.txt
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
out=gpurun_out/r06_c5_phase_cycles.txt
cat > $out <<'TXT'
phases of rollout_kernel_crowd<4> (experiment build -DSG_PHASE_TIMERS: a mark costs an s_memtime + a scalar add; shares, not absolute times)
 [7] run vote   [0] goal update, desire force   [6] crowd_pairs (the balanced pair loop)   [1] boundary terms, ped_move, controller, velocities, ego metrics
 [8] box centre / sin cos in fp32, cell index   [11] barrier before publishing   [2] publishing the tile's terms in LDS, stripe atomics   [12] range / dense vote (a barrier)
 [13] barrier behind the atomics   [9] stripe-mask reads + circle test per candidate   [10] the all-pairs walk instead (dense steps)
 [3] fp32 SAT filter   [15] fuzzy vote (a barrier)   [4] exact fp64 SAT, owner mapping   [5] statistics, events, terminal conditions, row stores, loop top
TXT
for wl in c5 c5roads; do
  echo "== $wl" >> $out
  SGYM_LIB=scenario_gym_amd/lib/ab/phase.so python3 bench.py --no-cpu-baseline --no-configs --workload $wl --steps 1 --warmup 0 2>&1 >/dev/null | grep "phase cycles" | tail -1 >> $out
done
cat $out
