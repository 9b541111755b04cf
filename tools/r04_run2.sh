#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "walker or crowd" 2>&1 | tail -15 > gpurun_out/r04b_walk_tests.txt; cat gpurun_out/r04b_walk_tests.txt
timeout 600 python3 bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04b_c5_bench.json 2> gpurun_out/r04b_c5_bench.err; cut -c1-300 gpurun_out/r04b_c5_bench.json; tail -3 gpurun_out/r04b_c5_bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04b_c5_trace -o t -- python3 bench.py --workload c5 --steps 1 --warmup 0 --no-cpu-baseline --verify 0 > gpurun_out/r04b_c5_trace.log 2>&1
f=$(find gpurun_out/r04b_c5_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r04b_c5_kernel_stats.csv && head -8 gpurun_out/r04b_c5_kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/r04b_c5_trace
