#!/bin/bash
# round 4, first GPU call: the GPU tests after the build split, the new bench line (c3, c5), phase timers of c5 / c3
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r04a_gputests.txt; cat gpurun_out/r04a_gputests.txt
python3 bench.py > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.err; cut -c1-600 gpurun_out/r04a_bench.json; tail -3 gpurun_out/r04a_bench.err
python3 bench.py --workload c5 --steps 5 --warmup 1 > gpurun_out/r04a_c5_bench.json 2> gpurun_out/r04a_c5_bench.err; cut -c1-400 gpurun_out/r04a_c5_bench.json
SGYM_LIB=scenario_gym_amd/lib/ab/phases.so python3 bench.py --workload c5 --steps 1 --warmup 0 --no-cpu-baseline --verify 0 > gpurun_out/r04a_c5_phases.json 2> gpurun_out/r04a_c5_phases.txt; grep -a "phase cycles" gpurun_out/r04a_c5_phases.txt | cut -c1-600
SGYM_LIB=scenario_gym_amd/lib/ab/phases.so python3 bench.py --workload c5 --sim-steps 1500 --steps 1 --warmup 0 --no-cpu-baseline --verify 0 > gpurun_out/r04a_c5_phases1500.json 2> gpurun_out/r04a_c5_phases1500.txt; grep -a "phase cycles" gpurun_out/r04a_c5_phases1500.txt | cut -c1-600
SGYM_LIB=scenario_gym_amd/lib/ab/phases.so python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --verify 0 > gpurun_out/r04a_c3_phases.json 2> gpurun_out/r04a_c3_phases.txt; grep -a "phase cycles" gpurun_out/r04a_c3_phases.txt | cut -c1-600
