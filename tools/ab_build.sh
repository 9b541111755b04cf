#!/bin/bash
# Experiment builds of libsgym_hip.so: tools/ab_build.sh <name> [-D... flags]  ->  scenario_gym_amd/lib/ab/<name>.so
# (select one at run time with SGYM_LIB=scenario_gym_amd/lib/ab/<name>.so; built .so files travel with gpurun)
name=$1; shift
cd "$(dirname "$0")/../scenario_gym_amd/csrc" && mkdir -p ../lib/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-array-bounds \
  -Wno-bitwise-instead-of-logical -Wno-unused-command-line-argument -mllvm --disable-promote-alloca-to-lds "$@" -shared -o ../lib/ab/$name.so sgym_hip.hip
