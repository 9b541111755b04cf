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

CATALOG = """<?xml version="1.0" encoding="utf-8"?>
<OpenSCENARIO><FileHeader description="synthetic" author="tools/ingest_rate.py" revMajor="1" revMinor="0" date="2026-01-01T00:00:00"/>
<Catalog name="SyntheticVehicleCatalog">
<Vehicle name="car1" vehicleCategory="car"><BoundingBox><Center x="1.37" y="0" z="0.8"/><Dimensions width="2.0" length="4.2" height="1.6"/></BoundingBox></Vehicle>
<Vehicle name="van" vehicleCategory="van"><BoundingBox><Center x="1.5" y="0" z="1.0"/><Dimensions width="2.2" length="5.6" height="2.2"/></BoundingBox></Vehicle>
</Catalog></OpenSCENARIO>
"""


def write_scenario(path, rng, n_entities, n_vertices):
    names = ["ego"] + [f"entity_{i}" for i in range(1, n_entities)]
    out = ['<?xml version="1.0" encoding="utf-8"?>\n<OpenSCENARIO>\n<FileHeader description="synthetic &amp; seeded" author="x" revMajor="1" '
           'revMinor="0" date="2026-01-01T00:00:00"/>\n<ParameterDeclarations/>\n<CatalogLocations><VehicleCatalog><Directory path="../Catalogs"/>'
           '</VehicleCatalog></CatalogLocations>\n<RoadNetwork/>\n<Entities>\n']
    for n in names:
        out.append(f'<ScenarioObject name="{n}"><CatalogReference catalogName="SyntheticVehicleCatalog" entryName="{"car1" if rng.random() < 0.8 else "van"}"/></ScenarioObject>\n')
    out.append("</Entities>\n<Storyboard>\n<Init><Actions>\n")
    starts = rng.uniform(-100, 100, (n_entities, 2))
    for n, (x, y) in zip(names, starts):
        out.append(f'<Private entityRef="{n}"><PrivateAction><TeleportAction><Position><WorldPosition x="{float(x)!r}" y="{float(y)!r}" z="0" h="0.5"/>'
                   "</Position></TeleportAction></PrivateAction></Private>\n")
    out.append("</Actions></Init>\n<Story name=\"s\"><Act name=\"a\">\n")
    for k, n in enumerate(names):
        t = np.linspace(0.0, 20.0, n_vertices) + (0.0 if k == 0 else rng.uniform(0, 2))
        h0, v = rng.uniform(-3, 3), rng.uniform(2, 12)
        xs, ys = starts[k, 0] + v * t * np.cos(h0), starts[k, 1] + v * t * np.sin(h0)
        out.append(f'<ManeuverGroup name="g{k}" maximumExecutionCount="1"><Actors selectTriggeringEntities="false"><EntityRef entityRef="{n}"/></Actors>'
                   f'<Maneuver name="m"><Event name="e" priority="overwrite"><Action name="act"><PrivateAction><RoutingAction><FollowTrajectoryAction>'
                   f'<Trajectory name="t" closed="false"><ParameterDeclarations/><Shape><Polyline>\n')
        for ti, x, y in zip(t, xs, ys):
            out.append(f'<Vertex time="{float(ti)!r}"><Position><WorldPosition x="{float(x)!r}" y="{float(y)!r}" h="{float(h0)!r}"/></Position></Vertex>\n')
        out.append("</Polyline></Shape></Trajectory><TimeReference><Timing domainAbsoluteRelative=\"absolute\" scale=\"1\" offset=\"0\"/></TimeReference>"
                   "<TrajectoryFollowingMode followingMode=\"position\"/></FollowTrajectoryAction></RoutingAction></PrivateAction></Action>"
                   "</Event></Maneuver></ManeuverGroup>\n")
    out.append("</Act></Story>\n<StopTrigger/>\n</Storyboard>\n</OpenSCENARIO>\n")
    with open(path, "w") as f:
        f.write("".join(out))


def make_directory(root, n_files, n_entities, n_vertices, seed=7):
    os.makedirs(os.path.join(root, "Catalogs"), exist_ok=True)
    os.makedirs(os.path.join(root, "Scenarios"), exist_ok=True)
    with open(os.path.join(root, "Catalogs", "catalog.xosc"), "w") as f:
        f.write(CATALOG)
    rng = np.random.default_rng(seed)
    paths = []
    for i in range(n_files):
        p = os.path.join(root, "Scenarios", f"s{i:05d}.xosc")
        write_scenario(p, rng, n_entities, n_vertices)
        paths.append(p)
    return paths


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
