#!/bin/bash
# which bench shapes read memory nobody wrote?  (SG_POISON fills unzeroed allocations)
for pz in 255 127; do
for args in "--scenarios 1024 --sim-steps 2000" "--scenarios 512 --sim-steps 2000" "--workload c3s --scenarios 512 --sim-steps 2000" "--workload c3s --scenarios 256 --sim-steps 2000" "--scenarios 4096 --sim-steps 2000"; do
  SG_POISON=$pz python3 bench.py $args --steps 2 --warmup 1 --verify 4 --no-cpu-baseline > /tmp/pb.out 2> /tmp/pb.err
  echo "poison $pz | $args | rc $? | $(grep -o 'DIFFERS.*' /tmp/pb.err | cut -c1-200) $(python3 -c "
import json
try:
    l=json.loads(open('/tmp/pb.out').read().strip().splitlines()[-1]); print(l['roofline']['kernel'], (l['roofline'].get('schedule') or {}).get('per_rank'), (l.get('verified') or {}).get('equal'))
except Exception as e: print('no line')
")"
done; done
