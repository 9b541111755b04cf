#!/usr/bin/env python3
"""The numbers DESIGN.md 6 / README / profiles/README quote, read from profiles/<tag>_*: one line per bench line of the tag, the
kernel-trace averages next to the HIP-event figures, the upload / RSS / timeline summaries.

    python tools/doc_numbers.py r03n
"""
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03n"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
for f in sorted(glob.glob(os.path.join(root, f"{tag}*_bench.json"))):
    lines = [x for x in open(f) if x.startswith("{")]
    if not lines:
        print(os.path.basename(f), "(no line)")
        continue
    l = json.loads(lines[-1])
    r = l.get("roofline") or {}
    s = r.get("valu_issue") or r.get("secondary") or {}
    hc = r.get("hbm_contract") or r
    val = f"{l['value'] / 1e9:.3f} G" if l["value"] > 1e6 else f"{l['value']:.1f}"
    print(f"{os.path.basename(f)[len(tag) + 1:]:28s} {val:>10s} {l['ms_per_step']:9.2f} ms  verified={(l.get('verified') or {}).get('equal')}  "
          f"{(r.get('kernel') or '')[4:]}  kernel_ms={r.get('kernel_ms') and round(r['kernel_ms'], 3)} gross={r.get('kernel_ms_gross') and round(r['kernel_ms_gross'], 3)} "
          f"launches={r.get('launches_per_rollout')} frac={r.get('frac') and round(r['frac'], 3)} traffic_ratio={hc.get('traffic_ratio') and round(hc['traffic_ratio'], 3)} bound={r.get('bound')} pipes={(r.get('pipelines') or {}).get('used_per_rank')} "
          f"valu_frac={s.get('frac') and round(s['frac'], 2)} sha={r.get('src_sha16')}")
    if "stage_seconds" in l:
        print("   ", l["stage_seconds"], "x_realtime", round(l.get("x_realtime", 0)), "cpu", (l.get("cpu_baseline") or {}).get("value"))
for name in (f"{tag}_kernel_stats.csv", f"{tag}_c3rss_kernel_stats.csv"):
    p = os.path.join(root, name)
    if os.path.exists(p):
        for r in list(csv.DictReader(open(p)))[:3]:
            print(f"{name}: {r['Name'][:64]}  calls={r['Calls']}  avg={float(r['AverageNs']) / 1e6:.3f} ms")
for name in (f"{tag}_upload_time.txt", f"{tag}_rss_time.txt", f"{tag}_timeline.txt"):
    p = os.path.join(root, name)
    if os.path.exists(p):
        print(name + ":", open(p).read().strip().split("\n")[-1])
for f in sorted(glob.glob(os.path.join(root, "latest_*_hbm_traffic.json")))[:1]:
    print("latest_* sha:", json.load(open(f)).get("src_sha16"))
