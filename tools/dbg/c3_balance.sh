#!/bin/bash
# The evidence behind "the rollout kernels and the controller chain are balanced" (HISTORY round 4): the c3 line of the
# product library, with s_setprio in the pre-pass (-DSG_CTL_SETPRIO), with the collision pass compiled out
# (tools/experiments/ablations.patch, -DSG_ABL_NO_COLL: WRONG results on purpose), and with both; two passes each, alternating;
# then the pre-pass launch durations of one pass of each from a kernel trace.  Build the libraries first:
#   git apply tools/experiments/ablations.patch; tools/ab_build.sh nocoll -DSG_ABL_NO_COLL; tools/ab_build.sh nocoll_prio -DSG_ABL_NO_COLL -DSG_CTL_SETPRIO
#   git apply -R tools/experiments/ablations.patch; tools/ab_build.sh prio -DSG_CTL_SETPRIO
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
val() { grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,1), 'G  kernel_ms', round(d['roofline']['kernel_ms'],3))"; }
for rep in 1 2; do
  for lib in "" prio nocoll nocoll_prio; do
    path=${lib:+scenario_gym_amd/lib/ab/$lib.so}
    echo -n "pass $rep  ${lib:-product}: "; SGYM_LIB=$path python3 bench.py --no-cpu-baseline --verify 0 --steps 4 --warmup 1 2>/dev/null | val
  done
done
for lib in "" nocoll nocoll_prio; do
  path=${lib:+scenario_gym_amd/lib/ab/$lib.so}
  SGYM_LIB=$path timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/bal_${lib:-product} -o t -- python3 bench.py --no-cpu-baseline --verify 0 --steps 1 --warmup 1 > /dev/null 2>&1
  python3 - "${lib:-product}" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
rows = []
for f in glob.glob(f"gpurun_out/bal_{tag}/**/*kernel_trace.csv", recursive=True)[:1]:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "rollout_kernel" in n or "control_kernel" in n:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "tab" if "rollout" in n else "ctl"))
rows.sort()
tabs = [r for r in rows if r[2] == "tab"]; n = len(tabs) // 2; t0 = tabs[-n][0]
last = [r for r in rows if r[0] >= t0]
ctl = [(r[1] - r[0]) / 1e6 for r in last if r[2] == "ctl"]
tab = [(r[1] - r[0]) / 1e6 for r in last if r[2] == "tab"]
gaps = []
cs = [r for r in last if r[2] == "ctl"]
for a, b in zip(cs, cs[1:]): gaps.append((b[0] - a[1]) / 1e6)
print(f"{tag}: last pass {(max(r[1] for r in last) - last[0][0]) / 1e6:.2f} ms; pre-pass launches {len(ctl)}, full-size ones {sorted(ctl)[-5:]} ms, idle between consecutive pre-pass launches {[round(g, 2) for g in gaps[-6:]]} ms; rollout launches {len(tab)}, median {sorted(tab)[len(tab) // 2]:.2f} ms")
PY
  rm -rf gpurun_out/bal_${lib:-product}
done
