#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "walker" 2>&1 | tail -15 > gpurun_out/r04b_walk_tests.txt; cat gpurun_out/r04b_walk_tests.txt
timeout 600 python3 bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04b_c5_bench.json 2> gpurun_out/r04b_c5_bench.err; cut -c1-300 gpurun_out/r04b_c5_bench.json; tail -3 gpurun_out/r04b_c5_bench.err
