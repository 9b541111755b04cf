#!/bin/bash
# Soak run on the GPU box: the randomized device-vs-oracle sweep with other seeds and more configurations.
# Usage: bash tools/fuzz.sh [n_configs] [seed ...]
n=${1:-150}; shift
seeds=${@:-"7 8 9"}
for s in $seeds; do
  SG_FUZZ_N=$n SG_FUZZ_SEED=$s python -m pytest tests/test_gpu_parity.py -q -x -k randomized_configurations 2>&1 | tail -2
done
# the social-force crowds (balanced pair loop, boundary terms), same idea
for s in $seeds; do
  SG_FUZZ_CROWDS=$n SG_FUZZ_SEED=$s python -m pytest tests/test_gpu_parity.py -q -x -k randomized_crowds 2>&1 | tail -2
done
# the RSS callback inside the rollout kernel
for s in $seeds; do
  SG_FUZZ_RSS=$n SG_FUZZ_SEED=$s python -m pytest tests/test_gpu_parity.py -q -x -k randomized_rss 2>&1 | tail -2
done
