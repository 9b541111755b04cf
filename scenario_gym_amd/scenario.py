"""Scenario container: the read API the rollout uses (reference scenario/scenario.py:20-100)."""
from typing import Dict, List, Optional

from .entity import Entity, Pedestrian, Vehicle
from .trajectory import Trajectory


class Scenario:
    def __init__(self, entities: List[Entity], name: Optional[str] = None, road_network=None, actions=None,
                 properties=None):
        self._entities = entities
        self._ref_to_entity: Dict[str, Entity] = {e.ref: e for e in entities}
        self.name = name
        self.road_network = road_network
        self.actions = actions if actions is not None else []
        self.properties = properties if properties is not None else {}

    @property
    def entities(self) -> List[Entity]:
        return self._entities

    def entity_by_name(self, e_ref: str) -> Optional[Entity]:
        return self._ref_to_entity.get(e_ref)

    @property
    def ego(self) -> Entity:
        """The entity with ref "ego", else the first entity (scenario.py:53-65)."""
        ego = self.entity_by_name("ego")
        return ego if ego is not None else self._entities[0]

    @property
    def vehicles(self):
        return [e for e in self._entities if isinstance(e, Vehicle)]

    @property
    def pedestrians(self):
        return [e for e in self._entities if isinstance(e, Pedestrian)]

    @property
    def trajectories(self) -> Dict[str, Trajectory]:
        return {e.ref: e.trajectory for e in self._entities}

    @property
    def length(self) -> float:
        """scenario.py:88-91."""
        return max(e.trajectory.max_t for e in self._entities)

    def translate(self, x):
        """scenario.py:157-177: every entity's trajectory translated by `x` ([t, x, y, z, h, p, r] offsets)."""
        new = self.copy()
        for e in new._entities:
            e.trajectory = e.trajectory.translate(x)
        new.name = self.name
        return new

    def reset_start(self, entity=None):
        """scenario.py:179-184: shift time so that `entity` (default: the ego) starts at t = 0."""
        import numpy as np

        start = (self.ego if entity is None else entity).trajectory.min_t
        return self.translate(np.array([-start, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]))

    def copy(self):
        return self.__class__([e.copy() for e in self._entities],
                              name=f"Copy of {self.name}" if self.name is not None else None,
                              road_network=self.road_network, actions=list(self.actions), properties=self.properties)
