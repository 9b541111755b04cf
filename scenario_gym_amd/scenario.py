"""Scenario container: the read API the rollout uses (reference scenario/scenario.py:20-100) and its JSON form
(scenario/scenario.py:186-319: from_dict / to_dict / from_json / to_json)."""
import json
import os
from typing import Any, Dict, List, Optional, Tuple, Type

from .entity import Entity, Pedestrian, Vehicle
from .trajectory import Trajectory


from .actions import (FixedTAction, ScenarioAction, ScenarioActionRecord, UpdateStateVariableAction,  # noqa: E402,F401
                      UserDefinedAction)


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

    def add_action(self, action, inplace: bool = False):
        """scenario.py:160-164."""
        scenario = self if inplace else self.copy()
        scenario.actions.append(action)
        return scenario

    def translate(self, x):
        """scenario.py:157-177: every entity's trajectory translated by `x` ([t, x, y, z, h, p, r] offsets)."""
        new = self.copy()
        for e in new._entities:
            e.trajectory = e.trajectory.translate(x)
        new.actions = [a.translate(x) for a in new.actions]  # FixedTAction.translate: t += x[0] (actions.py:102-106)
        new.name = self.name
        return new

    def reset_start(self, entity=None):
        """scenario.py:179-184: shift time so that `entity` (default: the ego) starts at t = 0."""
        import numpy as np

        start = (self.ego if entity is None else entity).trajectory.min_t
        return self.translate(np.array([-start, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]))

    # ------------------------------------------------------------------ JSON (scenario/scenario.py:186-319)
    @classmethod
    def from_dict(cls, data: Dict[str, Any], e_classes: Tuple[Type[Entity], ...] = (Vehicle, Pedestrian, Entity)):
        """Entities by their "entity_class" name among e_classes -- an unknown name (e.g. "MiscObject") falls to the LAST
        class of the tuple, as the reference's search loop leaves it; the road network from its file ("path", when it
        exists; else an empty network of that "name"; else none) or from the embedded dictionary."""
        from .road_network import RoadNetwork

        by_name = {c.__name__: c for c in e_classes}
        entities = [by_name.get(d.get("entity_class"), e_classes[-1]).from_dict(d) for d in data["entities"]]
        rn = data.get("road_network")
        if rn is not None:
            if rn.get("path") is not None:
                if os.path.exists(rn["path"]):
                    rn = RoadNetwork.create_from_file(str(rn["path"]))
                elif rn.get("name") is not None:
                    rn = RoadNetwork(name=rn["name"])
                else:
                    rn = None
            else:
                rn = RoadNetwork.create_from_dict(rn)
        # (scenario.py:214-221: "action_class" names one of a_classes, UpdateStateVariableAction by default)
        actions = [UpdateStateVariableAction.from_dict(a) for a in data.get("actions", ())]
        return cls(entities, name=data.get("name"), road_network=rn, actions=actions, properties=data.get("properties", {}))

    def to_dict(self, road_network_path: Optional[str] = "../Road_Networks") -> Dict[str, Any]:
        """road_network_path: a file, or a directory the network's "<name>.json" is looked for in (the default); None
        embeds the whole network."""
        if self.road_network is None:
            rn = None
        elif road_network_path is not None:
            p = road_network_path
            if not os.path.isfile(p):
                p = os.path.join(p, f"{self.road_network.name}.json")
            rn = {"path": p, "name": self.road_network.name}
        else:
            rn = self.road_network.to_dict()
        return {"entities": [e.to_dict() for e in self._entities], "name": self.name,
                "actions": [a.to_dict() for a in self.actions], "road_network": rn, "properties": self.properties}

    @classmethod
    def from_json(cls, path: str, road_network_dir: Optional[str] = None,
                  e_classes: Tuple[Type[Entity], ...] = (Vehicle, Pedestrian, Entity)):
        """A relative road-network path is taken from the file's directory, or from road_network_dir (itself relative to
        the file's directory unless absolute)."""
        with open(path) as f:
            data = json.load(f)
        rn = data.get("road_network")
        if rn is not None and rn.get("path") is not None and not os.path.isabs(rn["path"]):
            here = os.path.dirname(path)
            if road_network_dir is None:
                base = here
            elif os.path.isabs(road_network_dir):
                base = road_network_dir
            else:
                base = os.path.join(here, road_network_dir)
            rn["path"] = os.path.join(base, rn["path"])
        return cls.from_dict(data, e_classes=e_classes)

    def to_json(self, path: str, road_network_path: Optional[str] = "../Road_Networks") -> None:
        with open(path, "w") as f:
            json.dump(self.to_dict(road_network_path=road_network_path), f)

    def copy(self):
        return self.__class__([e.copy() for e in self._entities],
                              name=f"Copy of {self.name}" if self.name is not None else None,
                              road_network=self.road_network, actions=[a.copy() for a in self.actions], properties=self.properties)
