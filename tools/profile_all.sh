#!/bin/bash
# Usage (GPU box): bash tools/profile_all.sh <tag>   -- tools/profile_round.sh for every bench workload (c3 = <tag>_*, the others
# <tag>_<workload>_*), + the upload and RSS timings, the launch timeline, the chunk-launch (SG_QUEUE=0) and general-table-kernel lines,
# the per-item timeline of the persistent launch and the wide-scenario timings; then
# the counter passes go where bench.py looks for them (profiles/latest_* in the box's copy of the tree) and the bench lines are
# run again, so that they carry `traffic` and `secondary` of THESE sources.  ~17 minutes.  Copy gpurun_out/<tag>_* and
# gpurun_out/latest_* into profiles/ (not the *_trace / *_pmc_* directories).
tag=${1:-rXX}
bash tools/profile_round.sh ${tag} --steps 5 --warmup 1
bash tools/profile_round.sh ${tag}_c5 --workload c5 --steps 2 --warmup 1
bash tools/profile_round.sh ${tag}_c2 --workload c2 --steps 5 --warmup 1
bash tools/profile_round.sh ${tag}_c2s --workload c2s --steps 5 --warmup 1
bash tools/profile_round.sh ${tag}_c3rss --workload c3rss --steps 3 --warmup 1
bash tools/profile_round.sh ${tag}_c3s --workload c3s --steps 5 --warmup 1
bash tools/profile_round.sh ${tag}_c5mix --workload c5mix --steps 1 --warmup 1
bash tools/profile_round.sh ${tag}_c5roads --workload c5roads --steps 1 --warmup 1
SG_CROWD_ROADS=0 python3 bench.py --workload c5roads --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_c5roads_general_bench.json 2>/dev/null
SG_CROWD_RIDERS=0 python3 bench.py --workload c5mix --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_c5mix_general_bench.json 2>/dev/null
python3 bench.py --workload e2e > gpurun_out/${tag}_e2e_bench.json 2> gpurun_out/${tag}_e2e.err
for R in 512 1024 2048; do python3 bench.py --scenarios $R --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_shard_${R}_bench.json 2>/dev/null; python3 bench.py --workload c3s --scenarios $R --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_c3s_${R}_bench.json 2>/dev/null; done
python3 tools/upload_time.py > gpurun_out/${tag}_upload_time.txt 2>&1
python3 tools/rss_time.py > gpurun_out/${tag}_rss_time.txt 2>&1
python3 bench.py --workload c5 --ped-noise device --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_c5_noise_bench.json 2>/dev/null
bash tools/timeline.sh > gpurun_out/${tag}_timeline.txt 2>&1
cp gpurun_out/latest_* profiles/
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python3 bench.py --workload c5 --steps 2 --warmup 1 > gpurun_out/${tag}_c5_bench.json 2>/dev/null
python3 bench.py --workload c2 --steps 5 --warmup 1 > gpurun_out/${tag}_c2_bench.json 2>/dev/null
python3 bench.py --workload c2s --steps 5 --warmup 1 > gpurun_out/${tag}_c2s_bench.json 2>/dev/null
python3 bench.py --workload c3rss --steps 3 --warmup 1 > gpurun_out/${tag}_c3rss_bench.json 2>/dev/null
python3 bench.py --workload c3s --steps 5 --warmup 1 > gpurun_out/${tag}_c3s_bench.json 2>/dev/null
python3 bench.py --workload c5mix --steps 1 --warmup 1 > gpurun_out/${tag}_c5mix_bench.json 2>/dev/null
python3 bench.py --workload c5roads --steps 1 --warmup 1 > gpurun_out/${tag}_c5roads_bench.json 2>/dev/null
SG_PLANAR=0 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_nonplanar_bench.json 2>/dev/null
SG_QUEUE=0 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_chunk_launches_bench.json 2>/dev/null
python3 tools/dbg/queue_timeline.py > gpurun_out/${tag}_queue_timeline.txt 2>&1
python3 tools/dbg/wide_time.py > gpurun_out/${tag}_wide_time.txt 2>&1
tail -3 gpurun_out/${tag}_upload_time.txt gpurun_out/${tag}_rss_time.txt
