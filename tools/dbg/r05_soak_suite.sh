#!/bin/bash
# soak: the whole GPU suite N times in a row, stop at the first failure and keep its report
N=${1:-10}; shift
for i in $(seq 1 $N); do
  timeout 1300 python3 -m pytest tests -m gpu -x -q "$@" > /tmp/suite_$i.log 2>&1
  rc=$?
  echo "run $i: $(tail -1 /tmp/suite_$i.log)"
  if [ $rc -ne 0 ]; then grep -v Warning /tmp/suite_$i.log | tail -80; exit 1; fi
done
