cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
SGYM_LIB=$PWD/scenario_gym_amd/lib/ab/phases.so python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --verify 0 2>&1 | grep -E "phase cycles|value" | cut -c1-900 | tail -3
