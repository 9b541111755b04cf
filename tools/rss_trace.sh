#!/bin/bash
# Usage (GPU box): bash tools/rss_trace.sh  -- kernel trace of tools/rss_time.py (rollout_kernel_rss and rss_lines_kernel per launch)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
rm -rf gpurun_out/rss_trace
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rss_trace -o t -- python3 tools/rss_time.py > gpurun_out/rss_trace.log 2>&1
f=$(find gpurun_out/rss_trace -name "*kernel_stats.csv" | head -1)
head -8 "$f" | cut -c1-200
tail -2 gpurun_out/rss_trace.log
