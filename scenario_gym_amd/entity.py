"""Entity, BoundingBox, CatalogEntry: the parts of scenario_gym/entity/ and catalog_entry.py the
rollout path reads (reference entity/base.py:15-156, catalog_entry.py:83-138, 140-176).
XML / xosc (de)serialisation is out of scope."""
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
class CatalogEntry:
    """catalog_entry.py:140-176 (fields only)."""

    catalog: Optional[object]
    catalog_entry: str
    catalog_category: Optional[str]
    catalog_type: str
    bounding_box: BoundingBox
    properties: Dict[str, Union[float, str]] = field(default_factory=dict)
    files: List[str] = field(default_factory=list)


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
    pass


class Pedestrian(Entity):
    pass


class MiscObject(Entity):
    pass


def catalog_type_code(entity: Entity) -> int:
    """0 Vehicle, 1 Pedestrian, 2 other: what CollisionMetric.record_collision branches on
    (metrics/collision.py:85: `catalog_type != "Vehicle"` => non_vehicle)."""
    return {"Vehicle": 0, "Pedestrian": 1}.get(entity.catalog_entry.catalog_type, 2)
