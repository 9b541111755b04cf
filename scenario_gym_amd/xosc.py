"""Minimal OpenSCENARIO ingest for the rollout path (stdlib xml.etree only).

Reads what the device engine consumes -- entities, their catalog bounding boxes and their
trajectories -- from an .xosc file laid out like the reference's inputs:
  * CatalogLocations/*/Directory@path  -> *.xosc catalogs: Catalog/<Vehicle|Pedestrian|MiscObject>
    @name with BoundingBox/Center@x,y and BoundingBox/Dimensions@width,length
  * Entities/ScenarioObject@name with a CatalogReference@catalogName,@entryName
  * Storyboard/Init/Actions/Private@entityRef .../TeleportAction/Position/WorldPosition -> one knot at t=0
  * Storyboard/Story/Act/ManeuverGroup (Actors/EntityRef@entityRef) .../FollowTrajectoryAction/
    [TrajectoryRef/]Trajectory/Shape/Polyline/Vertex@time / Position/WorldPosition@x,y,z,h,p,r
Semantics follow the reference reader (xosc_interface/read.py:20-217, catalogs.py:30-84): missing
z/h/p/r become NaN and are filled by Trajectory.__init__, a later FollowTrajectoryAction replaces an
Init teleport, `relabel` renames entities to ego / vehicle_i / pedestrian_i / other_i.
Road networks, user-defined actions, properties and the writer are out of scope.
"""
import os
import warnings
import xml.etree.ElementTree as ET
from typing import Dict, Tuple

import numpy as np

from .road_network import RoadNetwork
from .entity import BoundingBox, CatalogEntry, Entity, MiscObject, Pedestrian, Vehicle
from .scenario import Scenario
from .trajectory import Trajectory

_ENTITY_CLASSES = {"Vehicle": Vehicle, "Pedestrian": Pedestrian, "MiscObject": MiscObject}
_catalog_cache: Dict[str, Tuple[str, Dict[str, Entity]]] = {}


def _entry_from_element(el, catalog_name):
    bb = el.find("BoundingBox")
    if bb is None:
        return None
    c, d = bb.find("Center"), bb.find("Dimensions")
    box = BoundingBox(float(d.attrib["width"]), float(d.attrib["length"]), float(c.attrib["x"]), float(c.attrib["y"]))
    cname = el.tag.lower() + "Category"
    ce = CatalogEntry(catalog_name, el.attrib["name"], el.attrib.get(cname), el.tag, box, {}, [])
    return _ENTITY_CLASSES.get(el.tag, Entity)(ce)


def read_catalog(catalog_file: str):
    """catalog name + {entry name: prototype Entity}."""
    if catalog_file not in _catalog_cache:
        root = ET.parse(catalog_file).getroot()
        cat = root.find("Catalog")
        entries = {}
        for el in list(cat):
            ent = _entry_from_element(el, cat.attrib["name"])
            if ent is not None:
                entries[ent.catalog_entry.catalog_entry] = ent
        _catalog_cache[catalog_file] = (cat.attrib["name"], entries)
    return _catalog_cache[catalog_file]


def _traj_point(t, wp):
    g = wp.attrib.get
    return np.array([t, float(wp.attrib["x"]), float(wp.attrib["y"]), float(g("z", np.nan)),
                     float(g("h", np.nan)), float(g("p", np.nan)), float(g("r", np.nan))])


def relabel_scenario(scenario: Scenario) -> Scenario:
    """read.py:244-273."""
    counts = {"vehicle": 0, "pedestrian": 0, "other": 0}
    scenario.entities[0].ref = "ego"
    for e in scenario.entities[1:]:
        key = "vehicle" if isinstance(e, Vehicle) else "pedestrian" if isinstance(e, Pedestrian) else "other"
        e.ref = f"{key}_{counts[key]}"
        counts[key] += 1
    scenario._ref_to_entity = {e.ref: e for e in scenario.entities}
    return scenario


def import_scenario(osc_file: str, relabel: bool = True) -> Scenario:
    if not os.path.exists(osc_file):
        raise FileNotFoundError(osc_file)
    cwd = os.path.dirname(osc_file)
    root = ET.parse(osc_file).getroot()

    catalogs: Dict[str, Dict[str, Entity]] = {}
    locs = root.find("CatalogLocations")
    for loc in ([] if locs is None else list(locs)):
        path = loc.find("Directory").attrib["path"]
        path = path if os.path.isabs(path) else os.path.join(cwd, path)
        for f in sorted(os.listdir(path)):
            if f.endswith(".xosc"):
                name, entries = read_catalog(os.path.join(path, f))
                catalogs[name] = entries

    entities: Dict[str, Entity] = {}
    for so in root.iterfind("Entities/ScenarioObject"):
        ref = so.attrib["name"]
        cat_ref = so.find("CatalogReference")
        ent = None
        if cat_ref is None:
            for el in list(so):  # inline Vehicle / Pedestrian / MiscObject definition
                ent = _entry_from_element(el, None) or ent
        else:
            try:
                ent = catalogs[cat_ref.attrib["catalogName"]][cat_ref.attrib["entryName"]].copy()
            except KeyError:
                warnings.warn(f"Could not find {cat_ref.attrib['entryName']} in catalog {cat_ref.attrib['catalogName']}")
        if ent is not None:
            ent.ref = ref
            entities[ref] = ent

    for private in root.iterfind("Storyboard/Init/Actions/Private"):
        ref = private.attrib["entityRef"]
        for wp in private.iterfind("PrivateAction/TeleportAction/Position/WorldPosition"):
            if ref in entities:
                entities[ref].trajectory = Trajectory(np.stack([_traj_point(0, wp)], axis=0))

    for mg in root.iterfind("Storyboard/Story/Act/ManeuverGroup"):
        er = mg.find("Actors/EntityRef")
        entity = entities.get(er.attrib["entityRef"]) if er is not None else None
        if entity is None:
            continue
        for event in mg.findall("Maneuver/Event"):
            fta = event.find("Action/PrivateAction/RoutingAction/FollowTrajectoryAction")
            if fta is None:
                continue
            verts = fta.findall("TrajectoryRef/Trajectory/Shape/Polyline/Vertex")
            verts += fta.findall("Trajectory/Shape/Polyline/Vertex")
            if verts:
                pts = [_traj_point(float(v.attrib["time"]), v.find("Position/WorldPosition")) for v in verts]
                entity.trajectory = Trajectory(np.stack(pts, axis=0))

    # road network: RoadNetwork/SceneGraphFile or LogicFile, ".json" when there is no extension (read.py:65-85)
    road_network = None
    rn = root.find("RoadNetwork/SceneGraphFile")
    if rn is None:
        rn = root.find("RoadNetwork/LogicFile")
    if rn is not None and rn.attrib.get("filepath"):
        path = rn.attrib["filepath"]
        path = path if os.path.isabs(path) else os.path.join(cwd, path)
        if os.path.splitext(path)[1] == "":
            path += ".json"
        if os.path.exists(path) and path.endswith(".json"):
            road_network = RoadNetwork.create_from_json(path)
    scenario = Scenario(list(entities.values()), name=os.path.splitext(os.path.basename(osc_file))[0],
                        road_network=road_network)
    return relabel_scenario(scenario) if relabel else scenario
