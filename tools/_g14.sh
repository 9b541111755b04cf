cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "crowd_riders" 2>&1 | tail -25
