#!/bin/bash
# per-step body time of the rollout items while the pre-pass runs and after, for the ways the pre-pass publishes its chunks
for wt in 0 2 4; do
  echo "=== uniform chunks of 1024, SG_QUEUE_CTLWT=$wt"
  SG_QUEUE_CTLWT=$wt SG_QUEUE_FIRST=1024 SG_QUEUE_CAP=1024 SG_QUEUE_DECAY=0 python3 tools/dbg/queue_timeline.py 2>/dev/null | grep -v "^t = " | cut -c1-150 | head -18
done
