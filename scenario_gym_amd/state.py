"""State: read-only view of one scenario of a device batch with the reference's State read API
(scenario_gym/state/state.py:20-395).  Values are fetched from HBM on demand and cached per step."""
from typing import Dict, List, Optional

import numpy as np

from .entity import Entity
from .scenario import Scenario


class State:
    def __init__(self, gym, index: int, scenario: Scenario, agents: dict, persist: bool):
        self._gym, self._i = gym, index
        self._scenario = scenario
        self.agents = agents
        self.persist = persist
        self.scenario_path: Optional[str] = None
        self.last_keystroke = None
        self.state_callbacks = []
        self._prev_poses_arr = None
        self._t_replay = None  # the clock value update_actions sees while a device rollout's actions are replayed
        self._reset_actions()

    # ------------------------------------------------------------------ scenario actions (state/state.py:150-160, 241-266)
    def _reset_actions(self) -> None:
        """State._reset_data: every action of the scenario is unapplied again, no entity has a state."""
        self.unapplied_actions = list(self._scenario.actions)
        self.action_apply_times = {a: float("nan") for a in self._scenario.actions}
        self.entity_state = dict.fromkeys(self._scenario.entities)

    def update_actions(self) -> None:
        """state.py:241-251: apply the actions whose trigger holds now, in the scenario's order; keep the others."""
        unapplied = []
        for act in self.unapplied_actions:
            if act.trigger_condition(self):
                self.apply_action(act)
                self.action_apply_times[act] = self.t
            else:
                unapplied.append(act)
        self.unapplied_actions = unapplied

    def apply_action(self, action) -> None:
        """state.py:253-262."""
        import warnings

        entity = self._scenario.entity_by_name(action.entity_ref)
        if entity is None:
            warnings.warn(f"No entity with name {action.entity_ref} was found for action {action.__class__.__name__}.")
        else:
            action.apply(self, entity)

    def _replay_actions(self, clock) -> None:
        """update_actions() once per value of `clock` (State.t after each step of a device call that ran several steps at
        once): what stepping one by one would have done, for actions whose trigger reads nothing but State.t."""
        if not self.unapplied_actions:
            return
        try:
            for t in clock:
                self._t_replay = float(t)
                self.update_actions()
                if not self.unapplied_actions:
                    break
        finally:
            self._t_replay = None

    # ------------------------------------------------------------------ plumbing
    def _s(self):
        return self._gym._fetch_state()

    def get_callback(self, cls):
        """state/state.py:272-282: the state callback of class `cls`, or None."""
        for cb in self._gym.state_callbacks:
            if isinstance(cb, cls):
                return cb
        return None

    @property
    def scenario(self) -> Scenario:
        return self._scenario

    @property
    def all_entities(self) -> List[Entity]:
        return list(self._scenario.entities)

    def _dict(self, arr) -> Dict[Entity, np.ndarray]:
        s = self._s()
        pres = s["present"][self._i]
        return {e: arr[self._i, k].copy() for k, e in enumerate(self._scenario.entities) if pres[k]}

    # ------------------------------------------------------------------ reference read API
    @property
    def t(self) -> float:
        if self._t_replay is not None:
            return self._t_replay
        return float(self._s()["t"][self._i])

    @property
    def prev_t(self) -> float:
        return float(self._s()["prev_t"][self._i])

    @property
    def dt(self) -> float:
        return self.t - self.prev_t

    @property
    def next_t(self) -> float:
        return self.t + self._gym.timestep

    @property
    def is_done(self) -> bool:
        return bool(self._s()["done"][self._i])

    def terminal_condition(self, name: str) -> bool:
        """TERMINAL_CONDITIONS[name](state) of the reference (state.py:397-408) on the current state, whatever the gym's own
        terminal_conditions are: all four are evaluated on the device for every scenario (sg_terminal_flags; ego_off_road
        against the scenario's road network -- none: off the road)."""
        from .engine import TERMINAL_BITS

        if name == "ego_off_road":
            self._gym._set_road_networks()
        return bool(self._gym._terminal_flags()[self._i] & TERMINAL_BITS[name])

    @property
    def poses(self) -> Dict[Entity, np.ndarray]:
        return self._dict(self._s()["poses"])

    @property
    def velocities(self) -> Dict[Entity, np.ndarray]:
        return self._dict(self._s()["vels"])

    @property
    def prev_poses(self) -> Dict[Entity, np.ndarray]:
        """poses - velocities * dt is NOT used: the previous step's poses are kept by the gym."""
        prev = self._gym._prev_state
        if prev is None:
            return {}
        now = self._s()["present"][self._i]
        return {e: prev["poses"][self._i, k].copy() for k, e in enumerate(self._scenario.entities)
                if now[k] and prev["present"][self._i, k]}

    @property
    def distances(self) -> Dict[Entity, float]:
        d = self._s()["dists"][self._i]
        return {e: float(d[k]) for k, e in enumerate(self._scenario.entities)}

    def collisions(self) -> Dict[Entity, List[Entity]]:
        """State.collisions() (state.py:306-310): adjacency rows computed on the device."""
        s = self._s()
        ents = self._scenario.entities
        rows, pres = s["coll"][self._i], s["present"][self._i]
        rows = rows.reshape(len(rows), -1)  # [E, words]: one 64-bit word per 64 entity slots (scenarios of any width)

        def listed(k):
            return [ents[j] for j in range(len(ents)) if (int(rows[k, j >> 6]) >> (j & 63)) & 1]

        return {e: listed(k) for k, e in enumerate(ents) if pres[k]}

    def recorded_poses(self, entity: Optional[Entity] = None):
        """state.py:272-290: (n, 7) rows [t, x, y, z, h, p, r] of the steps the entity was present."""
        t, poses = self._gym._fetch_record()
        ents = self._scenario.entities
        rows = int(self._s()["n_steps"][self._i]) + 1  # a scenario that ended before the batch did has no later rows

        def one(k):
            p = poses[:rows, self._i, k]
            ok = ~np.isnan(p[:, 0])
            return np.concatenate([t[:rows][ok, self._i, None], p[ok]], axis=1) if ok.any() else np.empty((0, 7))

        if entity is not None:
            return one(ents.index(entity))
        return {e: one(k) for k, e in enumerate(ents)}

    def get_entity_data(self, entity: Entity):
        """state.py:292-304."""
        return (self.t, self.next_t, self.poses.get(entity), self.velocities.get(entity),
                self.distances.get(entity), self.recorded_poses(entity=entity), self.entity_state.get(entity, None))

    def get_entity_box_points(self, e: Entity) -> np.ndarray:
        return e.get_bounding_box_points(self.poses[e])

    def future_collision(self, horizon: float = 5.0, n_samples: int = 10) -> bool:
        """FutureCollisionDetector(ego, horizon) at the current time (sensor/common.py:87-106), computed on the device."""
        return bool(self._gym._future(float(horizon), int(n_samples))[self._i])

    def entity_raster(self, width: float = 20.0, height: float = 20.0, nw: int = 20, nh: int = 20) -> np.ndarray:
        """RasterizedMapSensor "entity" layer around the ego (sensor/map.py:120-192), computed on the device: bool [nh, nw]."""
        return self._gym._raster(float(width), float(height), int(nw), int(nh))[self._i]

    def raster_map(self, layers, width: float = 20.0, height: float = 20.0, nw: int = 20, nh: int = 20) -> np.ndarray:
        """RasterizedMapSensor layers around the ego (sensor/map.py:136-271; names of its `_all_layers`), computed on the
        device from the scenario's road network: bool [n_layers, nh, nw]."""
        return self._gym._raster_map(tuple(layers), float(width), float(height), int(nw), int(nh))[self._i]

    def get_entities_in_area(self, area) -> List[Entity]:
        """state.py:340-354: entities whose centre point lies strictly inside `area`.  The reference takes a shapely
        (Multi)Polygon; here `area` is anything with `.exterior.coords`, or an (n, 2) array of ring vertices (simple
        polygon, either orientation; holes are not supported)."""
        ring = np.asarray(area.exterior.coords if hasattr(area, "exterior") else area, np.float64)[:, :2]
        if len(ring) > 1 and np.array_equal(ring[0], ring[-1]):
            ring = ring[:-1]
        ax, ay = ring[:, 0], ring[:, 1]
        bx, by = np.roll(ax, -1), np.roll(ay, -1)
        out = []
        for e, pose in self.poses.items():
            px, py = pose[0], pose[1]
            cr = (bx - ax) * (py - ay) - (by - ay) * (px - ax)
            on_edge = (cr == 0) & (np.minimum(ax, bx) <= px) & (px <= np.maximum(ax, bx)) & \
                      (np.minimum(ay, by) <= py) & (py <= np.maximum(ay, by))
            if on_edge.any():
                continue  # shapely `contains`: boundary points are outside
            cross = ((ay > py) != (by > py)) & (px < (bx - ax) * (py - ay) / np.where(by == ay, 1.0, by - ay) + ax)
            if cross.sum() % 2 == 1:
                out.append(e)
        return out

    def to_scenario(self, name: Optional[str] = None) -> Scenario:
        """state.py:374-394: a scenario whose trajectories are the recorded poses (stationary entities keep one
        knot).  Needs the gym's pose record (`record=True`, the default of ScenarioGym)."""
        from copy import deepcopy

        from .trajectory import Trajectory, is_stationary

        if name is None:
            name = f"Simulation of {self.scenario.name}" if self.scenario.name is None else None
        entities = []
        for entity, poses in self.recorded_poses().items():
            new_entity = deepcopy(entity)
            if is_stationary(poses):
                poses = poses[None, 0]
            new_entity.trajectory = Trajectory(poses)
            entities.append(new_entity)
        return Scenario(entities, name=name, road_network=self.scenario.road_network, actions=self.scenario.actions)

    def get_entities_in_radius(self, x: float, y: float, r: float) -> List[Entity]:
        """state.py:356-372.  The reference tests the centre against the 64-gon Point(x, y).buffer(r)."""
        ang = 2.0 * np.pi * np.arange(64) / 64
        vx, vy = x + r * np.cos(ang), y - r * np.sin(ang)
        out = []
        for e, pose in self.poses.items():
            ex, ey = np.roll(vx, -1) - vx, np.roll(vy, -1) - vy
            cr = ex * (pose[1] - vy) - ey * (pose[0] - vx)
            if (cr < 0).all():  # clockwise ring: strictly inside
                out.append(e)
        return out


# state.py:397-408: name -> predicate over a State, as the reference's callers index it (TERMINAL_CONDITIONS["collision"](state));
# ScenarioGym(terminal_conditions=[...]) takes the names (evaluated inside the rollout kernels) or any callable (host, per step).
TERMINAL_CONDITIONS = {name: (lambda s, _n=name: s.terminal_condition(_n))
                       for name in ("max_length", "collision", "ego_collision", "ego_off_road")}
