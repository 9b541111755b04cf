#!/bin/bash
# Experiment builds of libsgym_hip.so: tools/ab_build.sh <name> [-D... flags]  ->  scenario_gym_amd/lib/ab/<name>.so
# (select one at run time with SGYM_LIB=scenario_gym_amd/lib/ab/<name>.so; built .so files travel with gpurun)
name=$1; shift
cd "$(dirname "$0")/../scenario_gym_amd/csrc" && mkdir -p ../lib/ab
make -j8 -s OUT=../lib/ab/$name.d EXTRA="-Wno-unused-command-line-argument $*" ../lib/ab/$name.d/libsgym_hip.so && cp ../lib/ab/$name.d/libsgym_hip.so ../lib/ab/$name.so
