cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "randomized_configurations" 2>&1 | tail -12
(bash tools/fuzz.sh 400 11 12 13 14) > gpurun_out/r03_fuzz.txt 2>&1; tail -30 gpurun_out/r03_fuzz.txt
