"""Entity, BoundingBox, CatalogEntry: the parts of scenario_gym/entity/ and catalog_entry.py the
rollout path reads (reference entity/base.py:15-156, catalog_entry.py:83-138, 140-176).
JSON (de)serialisation as Scenario.to_json / from_json need it (scenario/scenario.py:186-319); writing xosc is out of scope."""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Union

import numpy as np

from .trajectory import Trajectory


@dataclass
class BoundingBox:
    """catalog_entry.py:83-91."""

    width: float
    length: float
    center_x: float
    center_y: float

    @classmethod
    def from_dict(cls, data):
        return cls(data["width"], data["length"], data["center_x"], data["center_y"])

    def to_dict(self):
        return dict(width=self.width, length=self.length, center_x=self.center_x, center_y=self.center_y)


@dataclass
class Catalog:
    """catalog_entry.py:11-25: the catalog an entry comes from (file-level name + the catalog group directory)."""

    name: str
    group_name: str

    @classmethod
    def from_dict(cls, data):
        return cls(data["name"], data["group_name"])

    def to_dict(self):
        return {"name": self.name, "group_name": self.group_name}


@dataclass
class Axle:
    """entity/vehicle.py:18-72."""

    max_steering: Optional[float]
    wheel_diameter: Optional[float]
    track_width: Optional[float]
    position_x: Optional[float]
    position_z: Optional[float]

    _KEYS = ("max_steering", "wheel_diameter", "track_width", "position_x", "position_z")

    @classmethod
    def from_dict(cls, data):
        return cls(*[data.get(k) for k in cls._KEYS])

    def to_dict(self):
        return {k: getattr(self, k) for k in self._KEYS}


@dataclass
class CatalogEntry:
    """catalog_entry.py:140-228.  `catalog` is a Catalog (or a bare catalog name, or None); the type-specific fields of the
    reference's VehicleCatalogEntry / PedestrianCatalogEntry / MiscObjectCatalogEntry (entity/vehicle.py:75-170,
    pedestrian.py:17-62, misc.py:17-60) live in `extra`, keyed as the reference's JSON keys them, in EXTRA_KEYS order."""

    catalog: Optional[object]
    catalog_entry: str
    catalog_category: Optional[str]
    catalog_type: str
    bounding_box: BoundingBox
    properties: Dict[str, Union[float, str]] = field(default_factory=dict)
    files: List[str] = field(default_factory=list)
    extra: Dict[str, object] = field(default_factory=dict)

    EXTRA_KEYS = {"Vehicle": ("mass", "max_speed", "max_deceleration", "max_acceleration", "front_axle", "rear_axle"),
                  "Pedestrian": ("mass",), "MiscObject": ("mass",)}

    def __getattr__(self, name):  # entry.mass, entry.front_axle, ... as the reference's subclasses have them
        extra = self.__dict__.get("extra")
        if extra is not None and name in extra:
            return extra[name]
        raise AttributeError(name)

    def to_dict(self, kind: Optional[str] = None):
        """JSON form; kind = the entity class the entry belongs to ("Vehicle", "Pedestrian", "MiscObject", else the base
        fields only), default: by catalog_type."""
        cat = self.catalog
        out = {"catalog": cat.to_dict() if isinstance(cat, Catalog) else (None if cat is None else {"name": cat, "group_name": None}),
               "catalog_entry": self.catalog_entry, "catalog_category": self.catalog_category,
               "catalog_type": self.catalog_type, "bounding_box": self.bounding_box.to_dict(),
               "properties": self.properties, "files": self.files}
        for k in self.EXTRA_KEYS.get(self.catalog_type if kind is None else kind, ()):
            v = self.extra.get(k)
            out[k] = v.to_dict() if isinstance(v, Axle) else v
        return out

    @classmethod
    def from_dict(cls, data, kind: Optional[str] = None):
        cat = data.get("catalog")
        extra = {}
        for k in cls.EXTRA_KEYS.get(data["catalog_type"] if kind is None else kind, ()):
            v = data.get(k)
            extra[k] = Axle.from_dict(v) if (k.endswith("_axle") and v is not None) else v
        return cls(Catalog.from_dict(cat) if cat is not None else None, data["catalog_entry"], data["catalog_category"],
                   data["catalog_type"], BoundingBox.from_dict(data["bounding_box"]), data.get("properties", {}),
                   data.get("files", []), extra)


class Entity:
    """entity/base.py:15-156."""

    def __init__(self, catalog_entry: CatalogEntry, trajectory: Optional[Trajectory] = None, ref: Optional[str] = None):
        self.ref = ref
        self.catalog_entry = catalog_entry
        self._trajectory = trajectory

    @property
    def trajectory(self) -> Trajectory:
        return self._trajectory

    @trajectory.setter
    def trajectory(self, trajectory: Trajectory) -> None:
        self._trajectory = trajectory

    @property
    def bounding_box(self) -> BoundingBox:
        return self.catalog_entry.bounding_box

    @property
    def type(self) -> Optional[str]:
        return self.catalog_entry.catalog_type.replace("Catalogs", "")

    def copy(self):
        return self.__class__(self.catalog_entry,
                              trajectory=None if self.trajectory is None else self.trajectory.copy(), ref=self.ref)

    __copy__ = copy

    # which reference entity class this is for the JSON form ("entity_class", entity/base.py:158-174)
    ENTRY_KIND: Optional[str] = None

    def to_dict(self):
        """entity/base.py:158-165."""
        return {"ref": self.ref, "trajectory": self.trajectory.to_json(),
                "catalog_entry": self.catalog_entry.to_dict(self.ENTRY_KIND or "Entity"), "entity_class": self.__class__.__name__}

    @classmethod
    def from_dict(cls, data):
        """entity/base.py:167-174."""
        return cls(CatalogEntry.from_dict(data["catalog_entry"], cls.ENTRY_KIND or "Entity"),
                   trajectory=Trajectory(np.array(data["trajectory"])), ref=data.get("ref"))

    def is_static(self) -> bool:
        return self.trajectory.data.shape[0] == 1

    def get_bounding_box_points(self, pose) -> np.ndarray:
        """Corners RR, FR, FL, RL in the global frame (entity/base.py:100-138); host-side helper."""
        pose = np.asarray(pose, np.float64)
        xy, h = pose[..., :2], pose[..., 3 if pose.shape[-1] > 3 else 2]
        b = self.bounding_box
        pts = np.array([
            [b.center_x - 0.5 * b.length, b.center_y + 0.5 * b.width],
            [b.center_x + 0.5 * b.length, b.center_y + 0.5 * b.width],
            [b.center_x + 0.5 * b.length, b.center_y - 0.5 * b.width],
            [b.center_x - 0.5 * b.length, b.center_y - 0.5 * b.width],
        ])
        c, s = np.cos(h)[..., None], np.sin(h)[..., None]
        x = pts[:, 0] * c + pts[:, 1] * (-s)
        y = pts[:, 0] * s + pts[:, 1] * c
        return xy[..., None, :] + np.stack([x, y], axis=-1)


class Vehicle(Entity):
    ENTRY_KIND = "Vehicle"


class Pedestrian(Entity):
    ENTRY_KIND = "Pedestrian"


class MiscObject(Entity):
    ENTRY_KIND = "MiscObject"


def catalog_type_code(entity: Entity) -> int:
    """0 Vehicle, 1 Pedestrian, 2 other: what CollisionMetric.record_collision branches on
    (metrics/collision.py:85: `catalog_type != "Vehicle"` => non_vehicle)."""
    return {"Vehicle": 0, "Pedestrian": 1}.get(entity.catalog_entry.catalog_type, 2)
