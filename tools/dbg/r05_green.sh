#!/bin/bash
# VERDICT r4 item 1 "done" evidence: the GPU suite three times in forward and three times in reversed file order + the schedule soak
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
tag=${1:-r05_box1}
mkdir -p gpurun_out
out=gpurun_out/${tag}_green.txt
echo "box: $(hostname) $(date -u +%FT%TZ) src $(python -c 'import scenario_gym_amd._lib as L; print(L.source_sha16())')" > $out
for i in 1 2 3; do
  timeout 1500 python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -1 | sed "s/^/forward $i: /" >> $out
  timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -1 | sed "s/^/reversed $i: /" >> $out
done
timeout 1500 python tools/schedule_soak.py 300 > gpurun_out/${tag}_schedule_soak.txt 2>&1
tail -1 gpurun_out/${tag}_schedule_soak.txt >> $out
cat $out
