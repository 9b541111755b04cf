#!/bin/bash
# the whole GPU suite (forward file order), log to gpurun_out/<tag>_gpu_tests.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
tag=${1:-r05}
mkdir -p gpurun_out
export SG_QUEUE_TIMEOUT_MS=${SG_QUEUE_TIMEOUT_MS:-8000}
timeout 3000 python -m pytest tests/ -q -m gpu -x --durations=15 ${@:2} > gpurun_out/${tag}_gpu_tests.txt 2>&1
tail -25 gpurun_out/${tag}_gpu_tests.txt
