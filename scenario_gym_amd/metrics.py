"""Metric API (reference metrics/base.py:8-73) with the built-in metrics as views over device results.

Built-ins (EgoAvgSpeed, EgoMaxSpeed, EgoDistanceTravelled, CollisionMetric) are accumulated inside the
rollout kernel; their `get_state()` reads the per-scenario row the gym fetched.  Any other Metric
subclass is stepped on the host after every device step (ScenarioGym falls back to one launch per step).
"""
from abc import ABC, abstractmethod
import numpy as np
from typing import Any, List, Optional, Tuple


class Metric(ABC):
    name: Optional[str] = None
    required_callbacks: List[type] = []
    device_field: Optional[str] = None  # set on built-ins: column of sg_metrics

    def __init__(self, name: Optional[str] = None):
        if name is not None:
            self.name = name
        elif self.name is None:
            self.name = self.__class__.__name__
        self.callbacks = []

    def reset(self, state) -> None:
        self._reset(state)

    def step(self, state) -> None:
        self._step(state)

    @abstractmethod
    def _reset(self, state) -> None:
        raise NotImplementedError

    @abstractmethod
    def _step(self, state) -> None:
        raise NotImplementedError

    @abstractmethod
    def get_state(self) -> Any:
        raise NotImplementedError


class _DeviceMetric(Metric):
    """A metric whose accumulator lives in sg_scenario_state."""

    def __init__(self, name: Optional[str] = None):
        super().__init__(name=name)
        self._value = None

    def _reset(self, state) -> None:
        self._value = None

    def _step(self, state) -> None:  # accumulated on the device
        pass

    def _load(self, row, events, refs) -> None:
        self._value = float(row[self.device_field])

    def get_state(self):
        return self._value


class EgoAvgSpeed(_DeviceMetric):
    """metrics/trajectory.py:8-28."""

    name = "ego_avg_speed"
    device_field = "ego_avg_speed"


class EgoMaxSpeed(_DeviceMetric):
    """metrics/trajectory.py:31-48."""

    name = "ego_max_speed"
    device_field = "ego_max_speed"


class EgoDistanceTravelled(_DeviceMetric):
    """metrics/trajectory.py:51-66."""

    name = "ego_distance_travelled"
    device_field = "ego_distance_travelled"


# CollisionTypes (metrics/collision.py:25-33); -1: a Vehicle hazard that is itself a controlled agent, left unclassified
COLLISION_TYPE_NAMES = {0: "other", 1: "t_bone", 2: "head_on", 3: "rear_end", 4: "side_swipe", 5: "non_vehicle", -1: "vehicle", -2: "vehicle"}


class CollisionMetric(_DeviceMetric):
    """metrics/collision.py:46-79: list of (t, other ref, type) for every NEW collision with the ego.

    `type` is "non_vehicle" for hazards whose catalog type is not "Vehicle", otherwise the class of
    record_collision (metrics/collision.py:81-203: "t_bone", "head_on", "rear_end", "side_swipe"), computed on the device
    with state.poses[...] where the reference reads the missing `Entity.pose` attribute (it raises there at this commit).
    A Vehicle hazard that is itself a controlled agent stays "vehicle" (unclassified)."""

    name = "collisions"
    device_field = "n_collisions"

    def __init__(self, c_tol: float = 0.4, name: Optional[str] = None):
        super().__init__(name=name)
        self.c_tol = c_tol
        self.collisions: List[Tuple[float, str, str]] = []

    def _reset(self, state) -> None:
        self.collisions = []

    def _load(self, row, events, refs) -> None:
        self.collisions = [(float(e["t"]), refs[int(e["other"])], COLLISION_TYPE_NAMES.get(int(e["type"]), "other"))
                           for e in events]

    def get_state(self):
        return list(self.collisions)


class CollisionPointMetric(_DeviceMetric):
    """metrics/collision.py:206-253: for every NEW collision with the ego (ref, collision point, relative heading): the
    centroid of the intersection of the two bounding boxes and (hazard heading - ego heading) mod 2 pi, computed on the
    device (sg_read_collision_points) with state.poses[...] where the reference reads the missing `Entity.pose`."""

    name = "collision_points"
    device_field = "n_collisions"

    def __init__(self, name: Optional[str] = None):
        super().__init__(name=name)
        self.collisions = []

    def _reset(self, state) -> None:
        self.collisions = []

    def _load(self, row, events, refs, points=None) -> None:
        self.collisions = [(refs[int(e["other"])], np.array([p[0], p[1]]), float(p[2])) for e, p in zip(events, points)]

    def get_state(self):
        return list(self.collisions)


class StateCallback(ABC):
    """callback.py:9-41: a callback that derives additional information from the state.  `reset(state)` resolves
    `required_callbacks` through `state.get_callback` (ValueError when one is missing) and calls `_reset`; the gym calls the
    object itself after the reset and after every step.  User subclasses run on the host (one launch per step)."""

    required_callbacks: List[type] = []

    def __init__(self):
        self.callbacks: list = []

    def reset(self, state) -> None:
        self.callbacks = []
        for req in self.required_callbacks:
            cb = state.get_callback(req)
            if cb is None:
                raise ValueError(f"Callback {req.__name__} is required for {self.__class__}.")
            self.callbacks.append(cb)
        self._reset(state)

    def _reset(self, state) -> None:
        pass

    @abstractmethod
    def __call__(self, state) -> None:
        raise NotImplementedError


class RSSDistances(StateCallback):
    """metrics/rss/callback.py:34-128 as a state callback: after every step the safe lateral / longitudinal distances
    between the ego and every present entity and the record the reference appends to that entity's history, computed on
    the device for the whole batch: a gym that holds this callback switches the library to run it after the reset and after
    every step of rollout / step (sg_set_rss), so it does not force the one-launch-per-step host path."""

    CODES = ("safe", "lateral", "longitudinal", "both", "unsafe_lateral", "unsafe_longitudinal", "found")

    def _reset(self, state) -> None:  # the library resets and updates the records itself (sg_set_rss)
        pass

    def __call__(self, state) -> None:
        pass

    def safe_distances(self, state):
        """{entity: [safe lateral, safe longitudinal]} of the latest update (entities it covered)."""
        _, _, codes, safe = state._gym._rss_results()
        ents = state.scenario.entities
        return {e: list(safe[state._i, k]) for k, e in enumerate(ents) if codes[state._i, k] >= 0}

    def latest_records(self, state):
        """{entity: record appended by the latest update} ("safe", "lateral", ..., "found")."""
        _, _, codes, _ = state._gym._rss_results()
        ents = state.scenario.entities
        return {e: self.CODES[codes[state._i, k]] for k, e in enumerate(ents) if codes[state._i, k] >= 0}


class RSS(_DeviceMetric):
    """metrics/rss/rss.py:106-163: {"safe_longitudinal": bool, "safe_lateral": bool} -- False once some entity's history
    holds the corresponding "unsafe_*" record.  Needs the RSSDistances callback (required_callbacks)."""

    name = "RSS"
    device_field = "n_steps"
    required_callbacks = [RSSDistances]

    def __init__(self, name: Optional[str] = None):
        super().__init__(name=name)
        self._flags = {"safe_longitudinal": True, "safe_lateral": True}

    def _reset(self, state) -> None:
        self._flags = {"safe_longitudinal": True, "safe_lateral": True}

    def _load(self, row, events, refs, flags=None) -> None:
        if flags is not None:
            self._flags = {"safe_longitudinal": bool(flags[0]), "safe_lateral": bool(flags[1])}

    def get_state(self):
        return dict(self._flags)
