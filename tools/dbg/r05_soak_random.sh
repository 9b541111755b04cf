#!/bin/bash
# soak: the randomized differential test N times in a row (one process each), stop at the first failure and keep its report
N=${1:-30}
for i in $(seq 1 $N); do
  timeout 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "test_randomized_configurations_match_oracle" > /tmp/soak_$i.log 2>&1
  rc=$?
  tail -1 /tmp/soak_$i.log
  if [ $rc -ne 0 ]; then grep -v Warning /tmp/soak_$i.log | tail -60; exit 1; fi
done
