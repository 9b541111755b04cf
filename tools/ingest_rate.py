#!/usr/bin/env python3
"""Ingest rate of a directory of OpenSCENARIO files: writes N synthetic .xosc files (the layout of the
reference's test inputs: a vehicle catalog, ScenarioObjects with CatalogReferences, Init teleports, one FollowTrajectoryAction
per entity) and times `import_scenario` -- the native scan (libsgym_xosc.so) against the ElementTree reader, serial, and
over a process pool.

    python tools/ingest_rate.py [n_files=10000] [entities=8] [vertices=120] [workers=the CPUs the cgroup quota grants]
"""
import os
import shutil
import sys
import tempfile
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from scenario_gym_amd.xosc_write import make_directory  # noqa: E402


def _load_native(p):
    from scenario_gym_amd.xosc import import_scenario
    return len(import_scenario(p).entities)


def _load_et(p):
    from scenario_gym_amd.xosc import import_scenario_et
    return len(import_scenario_et(p).entities)


def main():
    n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    V = int(sys.argv[3]) if len(sys.argv) > 3 else 120
    from scenario_gym_amd.packing import effective_cpus

    workers = int(sys.argv[4]) if len(sys.argv) > 4 else effective_cpus()
    root = tempfile.mkdtemp(prefix="sg_ingest_")
    try:
        t = time.perf_counter()
        paths = make_directory(root, n_files, E, V)
        mb = sum(os.path.getsize(p) for p in paths) / 1e6
        print(f"wrote {n_files} files x {E} entities x {V} vertices = {mb:.0f} MB in {time.perf_counter() - t:.1f} s")
        sample = paths[: min(300, n_files)]
        for name, fn in (("native scan (libsgym_xosc.so)", _load_native), ("ElementTree reader", _load_et)):
            fn(sample[0])
            t = time.perf_counter()
            for p in sample:
                fn(p)
            dt = time.perf_counter() - t
            print(f"{name}: {len(sample) / dt:.0f} files/s on one core ({mb / n_files * len(sample) / dt:.0f} MB/s)")
        t = time.perf_counter()
        with ProcessPoolExecutor(workers) as ex:
            n = sum(ex.map(_load_native, paths, chunksize=64))
        dt = time.perf_counter() - t
        print(f"native scan, {workers} processes: {n_files} files ({n} entities) in {dt:.2f} s = {n_files / dt:.0f} files/s")
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
