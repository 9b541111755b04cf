cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
(time timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "timed_shape or crowd_with_a_car" 2>&1 | tail -5) 2>&1 | tail -8
