#!/bin/bash
# Where do the persistent launch's 56 GB of WRITE_SIZE come from?  One rollout of c3 under --pmc WRITE_SIZE per variant.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
one() { # name, env...
  name=$1; shift
  rm -rf /tmp/w_$name
  ( export "$@" DUMMY=1; timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/w_$name -o p -- python3 bench.py --no-cpu-baseline --verify 0 --steps 1 --warmup 0 > /tmp/w_$name.log 2>&1 )
  python3 - $name <<'PY'
import csv, glob, sys, collections
n = sys.argv[1]
agg = collections.defaultdict(float)
for f in glob.glob(f"/tmp/w_{n}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "WRITE_SIZE":
            agg[r["Kernel_Name"].split("(")[0][-50:]] += float(r["Counter_Value"]) * 1024 / 1e9
print(n, {k: round(v, 2) for k, v in agg.items() if v > 0.05})
PY
}
one default
one noacq SG_QUEUE_ACQ=0
one nohandoff SG_QUEUE_HANDOFF=2
one neither SG_QUEUE_ACQ=0 SG_QUEUE_HANDOFF=2
one fence SG_QUEUE_HANDOFF=0
one chunks SG_QUEUE=0
