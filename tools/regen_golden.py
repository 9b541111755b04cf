#!/usr/bin/env python3
"""Regenerate every fixture under tests/golden/ from the real reference into a scratch directory and compare it, array by
array and byte for byte, with what is committed.  Build container only (needs /root/reference; the generators import it).

    python tools/regen_golden.py [--only roads,rss] [--out /tmp/regen] [--update]

Prints one line per generator (seconds, arrays identical / arrays) and the total "N / N"; exit code 1 when any array differs
or a committed array is missing from the regenerated file.  --update copies the regenerated files over the committed ones.
tests/test_oracle_golden.py::test_fixtures_regenerate_bit_for_bit runs the fast generators through this module.
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

# generator -> the files it writes
GENERATORS = {
    "make_golden": ["trajectory", "batch", "scenarios", "synth", "pid_xosc", "collision", "pedestrian"],
    "make_golden_actions": ["actions"],
    "make_golden_all_scenarios": ["all_scenarios"],
    "make_golden_collision_types": ["collision_types"],
    "make_golden_json": ["json", "elevation"],
    "make_golden_long": ["long"],
    "make_golden_mixed_peds": ["mixed_peds"],
    "make_golden_ped_noise": ["ped_noise"],
    "make_golden_ped_roads": ["ped_roads"],
    "make_golden_random_walk": ["random_walk"],
    "make_golden_roads": ["roads"],
    "make_golden_rss": ["rss"],
    "make_golden_sensors": ["sensors"],
}


def run_generator(name: str, out_dir: str) -> float:
    """One generator in a fresh interpreter with a random hash seed (a fixture must not depend on set / dict order)."""
    env = dict(os.environ, SG_GOLDEN_OUT=out_dir, PYTHONDONTWRITEBYTECODE="1", MPLBACKEND="Agg", PYTHONHASHSEED="random")
    t0 = time.time()
    subprocess.run([sys.executable, os.path.join(GOLD, name + ".py")], check=True, env=env, cwd=GOLD,
                   stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    return time.time() - t0


def compare_file(stem: str, out_dir: str):
    """(arrays identical, arrays committed, names that differ or are missing)."""
    gold = np.load(os.path.join(GOLD, stem + ".npz"), allow_pickle=False)
    got = np.load(os.path.join(out_dir, stem + ".npz"), allow_pickle=False)
    bad = []
    for k in gold.files:
        if k not in got.files:
            bad.append(k + " (missing)")
            continue
        a, b = gold[k], got[k]
        if a.dtype != b.dtype or a.shape != b.shape or a.tobytes() != b.tobytes():
            bad.append(k)
    bad += [k + " (new)" for k in got.files if k not in gold.files]
    return len(gold.files) - sum(1 for k in bad if not k.endswith("(new)")), len(gold.files), bad


def regenerate(names, out_dir, update=False, log=print):
    ok_total = n_total = 0
    failed = []
    for name in names:
        secs = run_generator(name, out_dir)
        for stem in GENERATORS[name]:
            ok, n, bad = compare_file(stem, out_dir)
            ok_total += ok
            n_total += n
            log(f"{name:32s} {stem:16s} {secs:6.1f} s  {ok} / {n}" + (f"  DIFFERENT: {bad[:6]}" if bad else ""))
            if bad:
                failed.append((stem, bad))
            if update:
                shutil.copyfile(os.path.join(out_dir, stem + ".npz"), os.path.join(GOLD, stem + ".npz"))
    log(f"total: {ok_total} / {n_total} arrays regenerate bit-identically")
    return ok_total, n_total, failed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--update", action="store_true")
    a = ap.parse_args()
    if not os.path.isdir("/root/reference/scenario_gym"):
        raise SystemExit("regen_golden: /root/reference is not here (build container only)")
    names = [n for n in GENERATORS if not a.only or n in {("make_golden_" + s) if s != "make_golden" else s for s in a.only.split(",")}]
    out_dir = a.out or tempfile.mkdtemp(prefix="regen_golden_")
    os.makedirs(out_dir, exist_ok=True)
    _, _, failed = regenerate(names, out_dir, update=a.update)
    raise SystemExit(1 if failed and not a.update else 0)


if __name__ == "__main__":
    main()
