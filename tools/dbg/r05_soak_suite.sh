#!/bin/bash
# soak: the whole GPU suite N times in a row (every failure reported, none stops the loop); the box's GPU id on top
N=${1:-10}; shift
rocminfo 2>/dev/null | tr -s " " | grep -i "Uuid: GPU"
for i in $(seq 1 $N); do
  timeout 1300 python3 -m pytest tests -m gpu -q "$@" > /tmp/suite_$i.log 2>&1
  echo "run $i: $(tail -1 /tmp/suite_$i.log)"
  grep "^FAILED" /tmp/suite_$i.log
  grep -v Warning /tmp/suite_$i.log | grep -B3 -A6 "^E  .*\(AssertionError\|Error\)" | cut -c1-600 | head -80
done
