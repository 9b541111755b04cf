cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
python3 bench.py > gpurun_out/r03k_bench.json 2> gpurun_out/r03k_bench.err
python3 bench.py --workload c5 --steps 2 --warmup 1 > gpurun_out/r03k_c5_bench.json 2>/dev/null
python3 bench.py --workload c2 --steps 5 --warmup 1 > gpurun_out/r03k_c2_bench.json 2>/dev/null
python3 bench.py --workload c2s --steps 5 --warmup 1 > gpurun_out/r03k_c2s_bench.json 2>/dev/null
python3 bench.py --workload c3rss --steps 3 --warmup 1 > gpurun_out/r03k_c3rss_bench.json 2>/dev/null
python3 bench.py --workload c3s --steps 5 --warmup 1 > gpurun_out/r03k_c3s_bench.json 2>/dev/null
python3 bench.py --workload c5mix --steps 1 --warmup 1 > gpurun_out/r03k_c5mix_bench.json 2>/dev/null
SG_PLANAR=0 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03k_nonplanar_bench.json 2>/dev/null
SG_TAB_SPLIT=1 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03k_one_pipeline_bench.json 2>/dev/null
cut -c1-300 gpurun_out/r03k_bench.json
