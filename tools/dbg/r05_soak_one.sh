#!/bin/bash
# soak one test expression N times; stop at the first failure and print its report: bash tools/dbg/r05_soak_one.sh "<-k expr>" N
expr=$1; N=${2:-30}
fails=0
for i in $(seq 1 $N); do
  timeout 300 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -k "$expr" > /tmp/one_$i.log 2>&1
  rc=$?
  echo "run $i: $(tail -1 /tmp/one_$i.log)"
  if [ $rc -ne 0 ]; then fails=$((fails+1)); grep -v Warning /tmp/one_$i.log | grep -B2 -A25 "^E  " | head -60; [ $fails -ge ${MAXFAIL:-1} ] && exit 1; fi
done
