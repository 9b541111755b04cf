"""Observations of the sensors (observation.py, sensor/common.py:54-58, 109-113, sensor/map.py:21-25): dataclasses with the
reference's field names, and `combine_observations`, which CombinedSensor uses to merge the observations of several sensors
of one entity into one class."""
import dataclasses
from dataclasses import dataclass
from typing import Any, Optional, Sequence

import numpy as np


@dataclass
class Observation:
    """observation.py:10-15: base class."""


@dataclass
class SingleEntityObservation(Observation):
    """observation.py:17-28: State.get_entity_data(entity) with the entity in front."""

    entity: Any
    t: float
    next_t: float
    pose: Optional[np.ndarray]
    velocity: Optional[np.ndarray]
    distance_travelled: Optional[float]
    recorded_poses: np.ndarray
    entity_state: Any


@dataclass
class FutureCollisionObservation(SingleEntityObservation):
    """sensor/common.py:54-58."""

    future_collision: bool


@dataclass
class CollisionObservation(SingleEntityObservation):
    """sensor/common.py:109-113."""

    collisions: dict


@dataclass
class MapObservation(SingleEntityObservation):
    """sensor/map.py:21-25."""

    map: np.ndarray


def combine_observations(*classes, prefixes: Optional[Sequence[Optional[str]]] = None):
    """observation.py:31-84: a dataclass holding the fields of all `classes` in order.  A field name that an earlier class
    already contributed is skipped -- or, with `prefixes` (one per class), taken as "<prefix>_<name>"; a name that is still
    taken then is an error.  The class has `from_obs(*observations)`, which builds it from one instance per input class."""
    if prefixes is not None and len(prefixes) != len(classes):
        raise ValueError("one prefix per observation class")
    fields, sources = [], []  # (name, type), (index of the class, its own field name)
    taken = set()
    for k, c in enumerate(classes):
        if not dataclasses.is_dataclass(c):
            raise TypeError(f"Observation {c} is not a dataclass.")
        for f in dataclasses.fields(c):
            name = f.name
            if name in taken:
                if prefixes is None:
                    continue
                name = f"{prefixes[k]}_{f.name}"
                if name in taken:
                    raise ValueError(f"Prefix {prefixes[k]} still leads to a duplicate name for {name}.")
            taken.add(name)
            fields.append((name, f.type))
            sources.append((k, f.name))

    def from_obs(cls, *obs):
        return cls(*(getattr(obs[k], name) for k, name in sources))

    return dataclasses.make_dataclass("CombinedObservation", fields, bases=(Observation,),
                                      namespace={"from_obs": classmethod(from_obs)})
