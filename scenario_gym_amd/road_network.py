"""Road networks: the polygons behind `RoadNetwork.driveable_surface` & co.

Host mirror of scenario_gym/road_network/{road_network,objects,base,utils}.py for what the device path consumes:
the JSON format (`Roads`, `Intersections`, optional `Lanes`, `Pavements`, `Crossings`, `Buildings`; each geometry a
`Boundary` ring -- or `{"exterior", "interiors"}` -- and for road-like objects a `Center` line), the object classes
with their driveable / walkable / impenetrable flags, and the unions the reference takes of them
(road_network.py:306-328, sensor/map.py:194-271).  No geometry is evaluated here: `polygon_arrays()` hands the rings
and their layer bits to the device (`sg_set_road_networks`), which answers `contains` for the `ego_off_road` terminal
condition and the RasterizedMapSensor layers.

Elevation: geometries may carry an `Elevation` list of (x, y, z) samples; `elevation_at_point` interpolates them as the
reference does (road_network.py:446-520) and the OpenSCENARIO reader uses it for trajectories without z.

Not mirrored: OpenDRIVE import, lane graphs, and the repair of invalid
(self-intersecting) boundaries through GEOS `make_valid` / `buffer` (base.py:94-118) -- rings are used as given, with
the even-odd rule.
"""
import json
import os
from functools import lru_cache
from typing import Any, Dict, List, Optional

import numpy as np

(LAYER_DRIVEABLE, LAYER_ROAD, LAYER_INTERSECTION, LAYER_LANE, LAYER_WALKABLE, LAYER_PAVEMENT, LAYER_CROSSING,
 LAYER_IMPENETRABLE) = (1 << i for i in range(8))

# RasterizedMapSensor._all_layers (sensor/map.py:44-53) -> layer code of sg_raster_map (0 = the entity layer)
LAYER_CODES = {"entity": 0, "driveable_surface": LAYER_DRIVEABLE, "road": LAYER_ROAD, "intersection": LAYER_INTERSECTION,
               "lane": LAYER_LANE, "walkable_surface": LAYER_WALKABLE, "pavement": LAYER_PAVEMENT, "crossing": LAYER_CROSSING}


def _ring(points) -> np.ndarray:
    r = np.array([[float(v["x"]), float(v["y"])] for v in points], np.float64).reshape(-1, 2)
    if len(r) > 1 and (r[0] == r[-1]).all():  # shapely closes rings itself; the device wants them open
        r = r[:-1]
    return r


def _key(data, name):
    return name if name in data else name.capitalize()


class RoadObject:
    """base.py:12-49."""

    def __init__(self, id: str):
        self.id = id

    def __eq__(self, other):
        if isinstance(other, str):
            return self.id == other
        return hasattr(other, "id") and other.id == self.id

    def __hash__(self):
        return hash(self.id)

    def __repr__(self):
        return f"{self.__class__.__name__}(id={self.id})"


class RoadGeometry(RoadObject):
    """base.py:52-126: an object with a boundary polygon.  boundary = exterior ring [n][2] (open), interiors = holes."""

    driveable = True
    walkable = True
    impenetrable = False

    has_center = False  # road-like objects (roads, lanes, pavements, crossings) write their centre line

    def __init__(self, id: str, boundary: np.ndarray, interiors: Optional[List[np.ndarray]] = None, center=None,
                 elevation=None):
        super().__init__(id)
        self.boundary = np.asarray(boundary, np.float64).reshape(-1, 2)
        self.interiors = [np.asarray(i, np.float64).reshape(-1, 2) for i in (interiors or [])]
        self.center = None if center is None else np.asarray(center, np.float64).reshape(-1, 2)
        if elevation is not None:
            elevation = np.asarray(elevation, np.float64)
            assert elevation.ndim == 2 and elevation.shape[1] == 3, "Invalid shape for elevation profile."
        self.elevation = elevation

    @classmethod
    def from_dict(cls, data: Dict[str, Any]):
        b = data["Boundary"]
        if isinstance(b, dict):
            ext, holes = _ring(b["exterior"]), [_ring(i) for i in b["interiors"]]
        elif isinstance(b, list):
            ext, holes = _ring(b), []
        else:
            raise ValueError(f"Type {type(b)} is not supported for boundary.")
        center = np.array([[v["x"], v["y"]] for v in data["Center"]], np.float64) if "Center" in data else None
        obj = cls(data["Id" if "Id" in data else "id"], ext, holes, center, data.get("Elevation"))
        obj._load_extra(data)
        return obj

    def _load_extra(self, data):
        pass

    def to_dict(self) -> Dict[str, Any]:
        """base.py:120-127, 159-165 + the per-class fields (objects.py:96-107, 137-141, 174-183, 225-229).  Rings are written
        closed (first vertex repeated), as shapely hands them out."""
        def pts(r):
            return [{"x": float(x), "y": float(y)} for x, y in np.concatenate([r, r[:1]])]

        out = {"id": self.id, "Boundary": pts(self.boundary) if not self.interiors else
               {"exterior": pts(self.boundary), "interiors": [pts(i) for i in self.interiors]},
               "Elevation": None if self.elevation is None else self.elevation.tolist()}
        if self.has_center:
            out["Center"] = [{"x": float(x), "y": float(y)} for x, y in (self.center if self.center is not None else [])]
        out.update(self._extra_dict())
        return out

    def _extra_dict(self) -> Dict[str, Any]:
        return {}

    def rings(self) -> List[np.ndarray]:
        return [self.boundary] + self.interiors


# objects.py:14-43: the lane types the reference's enum knows; anything else loads as "driving"
LANE_TYPES = ("driving", "none", "bidirectional", "biking", "border", "connectingRamp", "curb", "entry", "exit", "median",
              "mwyEntry", "mwyExit", "offRamp", "onRamp", "parking", "rail", "restricted", "roadWorks", "shoulder", "sidewalk",
              "special1", "special2", "special3", "stop", "taxi", "tram")


class Lane(RoadGeometry):
    walkable = False
    has_center = True

    def _load_extra(self, data):
        self.successors = list(set(data.get("successors", [])))
        self.predecessors = list(set(data.get("predecessors", [])))
        t = data.get("type", "driving")
        self.type = t if t in LANE_TYPES else "driving"

    def _extra_dict(self):
        return {"successors": self.successors, "predecessors": self.predecessors, "type": self.type}


class Road(RoadGeometry):
    walkable = False
    has_center = True

    def _load_extra(self, data):
        self.lanes = [Lane.from_dict(l) for l in data[_key(data, "lanes")]]

    def _extra_dict(self):
        return {"lanes": [l.to_dict() for l in self.lanes]}


class Intersection(RoadGeometry):
    driveable = True
    walkable = False

    def _load_extra(self, data):
        self.lanes = [Lane.from_dict(l) for l in data[_key(data, "lanes")]]
        self.connecting_roads = data.get("connecting_roads", [])

    def _extra_dict(self):
        return {"lanes": [l.to_dict() for l in self.lanes], "connecting_roads": self.connecting_roads}


class Pavement(RoadGeometry):
    driveable = False
    has_center = True


class Crossing(RoadGeometry):
    driveable = False
    has_center = True

    def _load_extra(self, data):
        self.pavements = data.get(_key(data, "pavements"), [])

    def _extra_dict(self):
        return {"pavements": self.pavements}


class Building(RoadGeometry):
    driveable = False
    impenetrable = True


class RoadNetwork:
    """road_network.py:29-328, the geometry side."""

    _default_object_names = {"roads": Road, "intersections": Intersection, "lanes": Lane, "pavements": Pavement,
                             "crossings": Crossing, "buildings": Building}

    def __init__(self, name: Optional[str] = None, properties=None, roads=(), intersections=(), lanes=(), pavements=(),
                 crossings=(), buildings=(), path: Optional[str] = None):
        self.name = name
        self.properties = properties if properties is not None else {}
        self.path = path
        self.roads, self.intersections = list(roads), list(intersections)
        self._lanes = list(lanes)
        self.pavements, self.crossings, self.buildings = list(pavements), list(crossings), list(buildings)
        self._arrays = None
        self._elev = None

    @classmethod
    def create_from_file(cls, filepath: str):
        if not os.path.exists(filepath):
            raise FileNotFoundError(f"File not found at: {os.path.abspath(filepath)}.")
        ext = os.path.splitext(filepath)[1]
        if ext in (".json", ""):
            return cls.create_from_json(filepath)
        raise ValueError(f"Unknown file type: {ext} (OpenDRIVE import is not part of this build).")

    @classmethod
    @lru_cache(maxsize=15)
    def create_from_json(cls, filepath: str):
        with open(filepath) as f:
            data = json.load(f)
        return cls.create_from_dict(data, name=os.path.splitext(os.path.basename(filepath))[0], path=filepath)

    @classmethod
    def create_from_dict(cls, data: Dict, **kwargs):
        assert "Roads" in data or "roads" in data, "Json data must contain road information."
        assert "Intersections" in data or "intersections" in data, "Json data must contain intersection information."
        objects = {}
        for obj, obj_cls in cls._default_object_names.items():
            key = obj if obj in data else (obj.capitalize() if obj.capitalize() in data else None)
            if key is not None:
                objects[obj] = [obj_cls.from_dict(d) for d in data[key]]
        if "name" not in kwargs and "name" in data:
            kwargs["name"] = data["name"]
        return cls(properties=data.get("properties"), **kwargs, **objects)

    @property
    def lanes(self) -> List[Lane]:
        """road_network.py:270-277: the lanes of every road and intersection, and the free-standing ones."""
        seen, out = set(), []
        for l in [l for x in self.roads + self.intersections for l in x.lanes] + self._lanes:
            if l.id not in seen:
                seen.add(l.id)
                out.append(l)
        return out

    @property
    def road_network_geometries(self) -> List[RoadGeometry]:
        return self.roads + self.intersections + self.lanes + self.pavements + self.crossings + self.buildings

    def to_dict(self) -> Dict[str, Any]:
        """road_network.py:409-414."""
        data = {"name": self.name, "properties": self.properties}
        for key, objs in (("roads", self.roads), ("intersections", self.intersections), ("lanes", self.lanes),
                          ("pavements", self.pavements), ("crossings", self.crossings), ("buildings", self.buildings)):
            data[key] = [o.to_dict() for o in objs]
        return data

    def to_json(self, filepath: str) -> None:
        with open(filepath, "w") as f:
            json.dump(self.to_dict(), f)

    # ------------------------------------------------------------------ elevation (road_network.py:446-520)
    def _elevation_model(self):
        """The (x, y, z) samples of every geometry that has them, thinned to at most ~5000 by a constant stride, triangulated
        once: (Delaunay, piecewise-linear interpolant on it, nearest-sample interpolant).  Without samples a flat unit
        square at z = 0 stands in.  (Order of the samples: roads, intersections, lanes, pavements, crossings, buildings, each
        in file order.  The reference collects its lanes through a Python set, i.e. in a per-process hash order, so with
        elevation on lanes its own sample order -- and with it the stride thinning -- varies from run to run.)"""
        if self._elev is None:
            from scipy.interpolate import LinearNDInterpolator, NearestNDInterpolator
            from scipy.spatial import Delaunay

            samples = [g.elevation for g in self.road_network_geometries if g.elevation is not None]
            pts = np.concatenate(samples, axis=0) if samples else np.array([[0, 1, 0], [1, 0, 0], [1, 1, 0], [0, 0, 0]])
            if pts.shape[0] > 5000:
                pts = pts[:: int(np.ceil(pts.shape[0] / 5000))]
            tri = Delaunay(pts[:, :2])
            self._elev = (tri, LinearNDInterpolator(pts[:, :2], pts[:, 2]), NearestNDInterpolator(pts[:, :2], pts[:, 2]))
        return self._elev

    def elevation_at_point(self, x, y) -> np.ndarray:
        """z at (x, y): linear on the triangle that contains the point, the nearest sample's z outside the samples' hull.
        x, y: scalars or 1-d arrays (a scalar is broadcast against an array)."""
        x, y = np.array(x), np.array(y)
        if x.ndim > 1 or y.ndim > 1:
            raise ValueError("x and y must be 0 or 1 dimensional.")
        both_1d = x.ndim == y.ndim == 1
        x, y = np.atleast_1d(x), np.atleast_1d(y)
        if x.shape[0] == 1 and y.shape[0] > 1:
            x = np.repeat(x, y.shape[0])
        elif y.shape[0] == 1 and x.shape[0] > 1:
            y = np.repeat(y, x.shape[0])
        tri, inside_fn, outside_fn = self._elevation_model()
        xy = np.column_stack((x, y))
        inside = tri.find_simplex(xy) >= 0
        z = np.empty(xy.shape[0])
        if inside.any():
            z[inside] = inside_fn(xy[inside])
        if (~inside).any():
            z[~inside] = outside_fn(xy[~inside])
        return z.squeeze() if both_1d else z

    def polygon_arrays(self) -> Dict[str, np.ndarray]:
        """Every boundary polygon once, with the unions it belongs to as LAYER_* bits:
        driveable / walkable / impenetrable surface = the geometries carrying that flag (road_network.py:306-328);
        road, intersection, pavement, crossing = that object list; lane = the lanes of the roads (sensor/map.py:236-243).
        Returns ring_off [P+1], vert_off [rings+1], verts [n][2], layers [P], ids [P]."""
        if self._arrays is not None:
            return self._arrays
        road_lane_ids = {l.id for r in self.roads for l in r.lanes}
        polys, layers, ids = [], [], []
        for kind, objs in (("road", self.roads), ("intersection", self.intersections), ("lane", self.lanes),
                           ("pavement", self.pavements), ("crossing", self.crossings), ("building", self.buildings)):
            for g in objs:
                bits = (LAYER_DRIVEABLE if g.driveable else 0) | (LAYER_WALKABLE if g.walkable else 0) | \
                       (LAYER_IMPENETRABLE if g.impenetrable else 0)
                bits |= {"road": LAYER_ROAD, "intersection": LAYER_INTERSECTION, "pavement": LAYER_PAVEMENT,
                         "crossing": LAYER_CROSSING}.get(kind, 0)
                if kind == "lane" and g.id in road_lane_ids:
                    bits |= LAYER_LANE
                polys.append(g.rings())
                layers.append(bits)
                ids.append(g.id)
        ring_off = np.concatenate([[0], np.cumsum([len(r) for r in polys])]).astype(np.int64)
        rings = [r for p in polys for r in p]
        vert_off = np.concatenate([[0], np.cumsum([len(r) for r in rings])]).astype(np.int64)
        verts = np.concatenate(rings, axis=0) if rings else np.zeros((0, 2))
        self._arrays = dict(ring_off=ring_off, vert_off=vert_off, verts=np.ascontiguousarray(verts, np.float64),
                            layers=np.array(layers, np.uint32), ids=np.array(ids))
        return self._arrays
