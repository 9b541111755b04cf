"""A minimal OpenSCENARIO writer: what the importer (xosc.py / libsgym_xosc.so) reads back -- catalog references, Init
teleports, one FollowTrajectoryAction per entity -- as plain text.  Used to generate synthetic file sets (bench.py
--workload e2e, tools/ingest_rate.py) and test inputs; the reference's scenariogeneration-based writer
(xosc_interface/write.py) is out of scope."""
import os
from typing import Iterable, Optional, Sequence, Tuple

import numpy as np

CATALOG_TEXT = """<?xml version="1.0" encoding="utf-8"?>
<OpenSCENARIO><FileHeader description="synthetic" author="scenario_gym_amd.xosc_write" revMajor="1" revMinor="0" date="2026-01-01T00:00:00"/>
<Catalog name="SyntheticVehicleCatalog">
<Vehicle name="car1" vehicleCategory="car"><BoundingBox><Center x="1.37" y="0" z="0.8"/><Dimensions width="2.0" length="4.2" height="1.6"/></BoundingBox></Vehicle>
<Vehicle name="van" vehicleCategory="van"><BoundingBox><Center x="1.5" y="0" z="1.0"/><Dimensions width="2.2" length="5.6" height="2.2"/></BoundingBox></Vehicle>
</Catalog></OpenSCENARIO>
"""
CATALOG_NAME = "SyntheticVehicleCatalog"


def write_catalog(directory: str) -> str:
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, "catalog.xosc")
    with open(path, "w") as f:
        f.write(CATALOG_TEXT)
    return path


def _wp(row, fields="xyzhpr") -> str:
    """WorldPosition attributes of a knot row [t, x, y, z, h, p, r]; NaN = attribute absent."""
    return " ".join(f'{k}="{float(v)!r}"' for k, v in zip(fields, row[1:7]) if v == v)


def write_scenario(path: str, entities: Sequence[Tuple[str, str, np.ndarray]], catalog_dir: str = "../Catalogs",
                   road_network_file: Optional[str] = None, description: str = "synthetic") -> None:
    """entities: (name, catalog entry, knots [n, 7] = t, x, y, z, h, p, r with NaN for attributes to leave out).  Every entity
    gets an Init teleport to its first knot and, with more than one knot, a FollowTrajectoryAction over all of them."""
    out = ['<?xml version="1.0" encoding="utf-8"?>\n<OpenSCENARIO>\n'
           f'<FileHeader description="{description}" author="x" revMajor="1" revMinor="0" date="2026-01-01T00:00:00"/>\n'
           f'<ParameterDeclarations/>\n<CatalogLocations><VehicleCatalog><Directory path="{catalog_dir}"/></VehicleCatalog></CatalogLocations>\n']
    out.append("<RoadNetwork/>\n" if road_network_file is None else
               f'<RoadNetwork><LogicFile filepath="{road_network_file}"/><SceneGraphFile filepath="{road_network_file}"/></RoadNetwork>\n')
    out.append("<Entities>\n")
    for name, entry, _ in entities:
        out.append(f'<ScenarioObject name="{name}"><CatalogReference catalogName="{CATALOG_NAME}" entryName="{entry}"/></ScenarioObject>\n')
    out.append("</Entities>\n<Storyboard>\n<Init><Actions>\n")
    for name, _, knots in entities:
        out.append(f'<Private entityRef="{name}"><PrivateAction><TeleportAction><Position><WorldPosition {_wp(knots[0])}/>'
                   "</Position></TeleportAction></PrivateAction></Private>\n")
    out.append('</Actions></Init>\n<Story name="s"><Act name="a">\n')
    for k, (name, _, knots) in enumerate(entities):
        if len(knots) < 2:
            continue
        out.append(f'<ManeuverGroup name="g{k}" maximumExecutionCount="1"><Actors selectTriggeringEntities="false"><EntityRef entityRef="{name}"/></Actors>'
                   '<Maneuver name="m"><Event name="e" priority="overwrite"><Action name="act"><PrivateAction><RoutingAction><FollowTrajectoryAction>'
                   '<Trajectory name="t" closed="false"><ParameterDeclarations/><Shape><Polyline>\n')
        for row in knots:
            out.append(f'<Vertex time="{float(row[0])!r}"><Position><WorldPosition {_wp(row)}/></Position></Vertex>\n')
        out.append('</Polyline></Shape></Trajectory><TimeReference><Timing domainAbsoluteRelative="absolute" scale="1" offset="0"/></TimeReference>'
                   '<TrajectoryFollowingMode followingMode="position"/></FollowTrajectoryAction></RoutingAction></PrivateAction></Action>'
                   "</Event></Maneuver></ManeuverGroup>\n")
    out.append("</Act></Story>\n<StopTrigger/>\n</Storyboard>\n</OpenSCENARIO>\n")
    with open(path, "w") as f:
        f.write("".join(out))


def synthetic_entities(rng, n_entities: int, n_vertices: int, duration: float = 20.0, extent: float = 100.0):
    """Straight constant-speed tracks (x, y, h given; z, p, r left out), the first one ("ego") from t = 0."""
    names = ["ego"] + [f"entity_{i}" for i in range(1, n_entities)]
    starts = rng.uniform(-extent, extent, (n_entities, 2))
    ents = []
    for k, n in enumerate(names):
        t = np.linspace(0.0, duration, n_vertices) + (0.0 if k == 0 else rng.uniform(0, duration / 10))
        h0, v = rng.uniform(-3, 3), rng.uniform(2, 12)
        knots = np.full((n_vertices, 7), np.nan)
        knots[:, 0], knots[:, 4] = t, h0
        knots[:, 1], knots[:, 2] = starts[k, 0] + v * t * np.cos(h0), starts[k, 1] + v * t * np.sin(h0)
        ents.append((n, "car1" if rng.random() < 0.8 else "van", knots))
    return ents


def make_directory(root: str, n_files: int, n_entities: int, n_vertices: int, seed: int = 7, duration: float = 20.0,
                   extent: float = 100.0) -> Iterable[str]:
    """root/Catalogs/catalog.xosc + root/Scenarios/s00000.xosc ...; returns the scenario paths."""
    write_catalog(os.path.join(root, "Catalogs"))
    os.makedirs(os.path.join(root, "Scenarios"), exist_ok=True)
    rng = np.random.default_rng(seed)
    paths = []
    for i in range(n_files):
        p = os.path.join(root, "Scenarios", f"s{i:05d}.xosc")
        write_scenario(p, synthetic_entities(rng, n_entities, n_vertices, duration, extent), description="synthetic &amp; seeded")
        paths.append(p)
    return paths
