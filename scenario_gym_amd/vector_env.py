"""A vector of scenario_gym RL environments on one device.

The reference's `integrations/openaigym.py` wraps ONE ScenarioGym as a gym `Env`: `step(action)` runs the ego's
VehicleController with the policy's (acceleration, steering), every other entity by its trajectory, then returns the
ego's rasterized-map observation, the agent's reward and `state.is_done` (:171-226); the defaults are
terminal_conditions ["max_length", "ego_collision", "ego_off_road"] (:93-94), `VehicleController(max_steer=0.9,
max_accel=5.0)` and `MapOnlySensor(channels_first=True, height=30, width=30, n=128)` with the default layers
(entity, driveable_surface) (:280-293), reward -1 for a done state that is off the road or in an ego collision and 0.01
otherwise (:300-310).  `VectorScenarioEnv` is that loop for R scenarios at once: one `sg_tick` per tick (the step with the
[R, 2] actions, the terminal conditions and the map observation replayed as one captured hipGraph) and
`sg_reset_scenarios` for the environments whose episode ended.  No arithmetic of the environment runs on the host.
"""
from typing import Optional, Sequence

import numpy as np

from . import _lib as L
from .agent import ExternalVehicleAgent
from .engine import RolloutEngine
from .packing import pack_scenarios
from .road_network import LAYER_CODES
from .scenario import Scenario


class VectorScenarioEnv:
    def __init__(self, scenarios: Sequence[Scenario], timestep: float = 0.1,
                 terminal_conditions: Optional[Sequence[str]] = None, layers: Optional[Sequence[str]] = None,
                 height: float = 30.0, width: float = 30.0, n: int = 128, max_steer: float = 0.9, max_accel: float = 5.0,
                 auto_reset: bool = True, torch_obs: bool = False, device: int = 0):
        self.scenarios = list(scenarios)
        self.terminal_conditions = list(terminal_conditions) if terminal_conditions is not None else \
            ["max_length", "ego_collision", "ego_off_road"]
        self.layers = list(layers) if layers is not None else ["entity", "driveable_surface"]
        self._codes = [LAYER_CODES[l] for l in self.layers]
        self.height, self.width, self.n = float(height), float(width), int(n)
        self.auto_reset, self.torch_obs = auto_reset, torch_obs

        def create_agent(scenario, entity):  # openaigym.py:280-293: the ego is driven by the policy
            if entity is scenario.ego:
                return ExternalVehicleAgent(entity, max_steer=max_steer, max_accel=max_accel)

        packed, _ = pack_scenarios(self.scenarios, create_agent)
        self.n_envs = packed.n_scenarios
        self.engine = RolloutEngine(packed.n_scenarios, packed.n_entities, timestep=timestep,
                                    terminal_conditions=self.terminal_conditions, device=device)
        self.engine.upload(packed)
        nets, index, net_of = [], {}, []
        for sc in self.scenarios:  # shared road networks go down once
            rn = sc.road_network
            if rn is not None and id(rn) not in index:
                index[id(rn)] = len(nets)
                nets.append(rn.polygon_arrays())
            net_of.append(-1 if rn is None else index[id(rn)])
        self.engine.set_road_networks(nets, net_of)
        self._mask = 0
        for c in self.terminal_conditions:
            self._mask |= {"max_length": L.TERM_MAX_LENGTH, "collision": L.TERM_COLLISION,
                           "ego_collision": L.TERM_EGO_COLLISION, "ego_off_road": L.TERM_EGO_OFF_ROAD}[c]
        self.done = np.zeros(self.n_envs, bool)

    @property
    def observation_shape(self):
        return (len(self.layers), self.n, self.n)  # channels first, as MapOnlySensor(channels_first=True)

    def _observe(self):
        if self.torch_obs:
            return self.engine.raster_map_torch(self._codes, self.width, self.height, self.n, self.n)
        return self.engine.raster_map(self._codes, self.width, self.height, self.n, self.n)

    def reset(self):
        """Env.reset for every environment (openaigym.py:128-169): observations [R, n_layers, n, n]."""
        self.engine.reset()
        self.done[:] = False
        return self._observe()

    def step(self, actions):
        """actions [R, 2] = (acceleration, steering) per environment (numpy, or a float64 torch tensor on the device).
        Returns (obs, reward [R], done [R], info).  With auto_reset the environments that finished are reset and their
        observation is the first of the new episode; without it, stepping a finished environment raises as the reference's
        `step` does."""
        if not self.auto_reset and self.done.any():
            raise ValueError("Step called when state is terminal.")
        if not hasattr(actions, "data_ptr"):
            actions = np.asarray(actions, np.float64).reshape(self.n_envs, 2)
        # step + terminal conditions + observation: one captured graph launch (sg_tick)
        obs, flags = self.engine.tick(actions, self._codes, self.width, self.height, self.n, self.n, torch_out=self.torch_obs)
        if self.torch_obs:
            flags = flags.cpu().numpy().astype(np.uint32)
        done = (flags & self._mask) != 0
        bad = (flags & (L.TERM_EGO_OFF_ROAD | L.TERM_EGO_COLLISION)) != 0
        reward = np.where(done & bad, -1.0, 0.01)  # RLAgent.reward, openaigym.py:300-310
        self.done = done
        if self.auto_reset and done.any():
            self.engine.reset_scenarios(done)
            self.done = np.zeros(self.n_envs, bool)
            obs = self._observe()  # the restarted environments return the first observation of their new episode
        return obs, reward, done, {"terminal_flags": flags}

    def close(self):
        self.engine.close()
