#!/bin/bash
# two live ranks on one GPU (the command of test_bench_two_live_ranks_on_one_gpu), allocations poisoned; variants by environment
run() { # name env...
  name=$1; shift
  for k in 1 2 3; do
    ( export "$@" SG_POISON=${PZ:-255} SGYM_DIST_BACKEND=gloo SGYM_DIST_ONE_DEVICE=1
      python3 bench.py --gpus 2 --scenarios 1024 --sim-steps 2000 --steps 2 --warmup 1 --verify 4 > /tmp/t1.out 2> /tmp/t1.err; rc=$?
      echo "$name run $k: rc $rc | $(grep -ho 'DIFFERS.\{0,200\}\|gave up.\{0,80\}' /tmp/t1.err | head -2 | tr '\n' ' ')" )
  done
}
run queue A=1
run chunk_launches SG_QUEUE=0
run no_poison SG_POISON=0
