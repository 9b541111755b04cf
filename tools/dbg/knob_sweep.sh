#!/bin/bash
# env knobs of the launch schedules against their defaults (c3, c3s, c2s, c3rss)
cd "$GRAFT_REPO_ROOT"
val() { grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,2))"; }
run() { python3 bench.py --no-cpu-baseline --verify 0 "$@" 2>/dev/null | val; }
echo -n "c3 default: "; run --steps 6 --warmup 1
for c in 512 768 1536 2048; do echo -n "c3 SG_CHUNK_STEPS=$c: "; SG_CHUNK_STEPS=$c run --steps 6 --warmup 1; done
echo -n "c3 default again: "; run --steps 6 --warmup 1
echo -n "c3s default: "; run --workload c3s --steps 6 --warmup 1
for l in 96 128 224 320; do echo -n "c3s SG_SLICE_LEN=$l: "; SG_SLICE_LEN=$l run --workload c3s --steps 6 --warmup 1; done
for l in 1024 4096; do echo -n "c3s SG_SLICE_CTL_STEPS=$l: "; SG_SLICE_CTL_STEPS=$l run --workload c3s --steps 6 --warmup 1; done
echo -n "c2s default: "; run --workload c2s --steps 6 --warmup 1
for l in 96 128 224 320; do echo -n "c2s SG_SLICE_LEN=$l: "; SG_SLICE_LEN=$l run --workload c2s --steps 6 --warmup 1; done
echo -n "c3rss default: "; run --workload c3rss --steps 3 --warmup 1
for l in 512 2048; do echo -n "c3rss SG_RSSQ_STEPS=$l: "; SG_RSSQ_STEPS=$l run --workload c3rss --steps 3 --warmup 1; done
