"""Agents, controllers, sensors, actions: the reference's plugin classes.

In the reference these objects compute one pose per step in Python (agent.py:52-57,
controller.py:30-42).  Here the built-in kinds are recognised by type when a scenario is packed
and lowered to device lanes (SG_KIND_*): for them the objects only carry their parameters.  Any OTHER
`Agent` subclass (a Python `_step`) keeps the reference protocol -- sensor.step -> _step -> controller.step
-- on the host, once per tick, and its pose is injected into the device step (SG_KIND_AGENT_EXTERNAL).
Class names, constructor signatures and defaults follow the reference (agent.py:18-169,
controller.py:12-258, action.py:12-83, sensor/base.py:9-54, sensor/common.py:39-50).
"""
from typing import Optional

import numpy as np

from . import _lib as L
from .entity import Entity
from .observation import (CollisionObservation, FutureCollisionObservation, MapObservation, SingleEntityObservation,
                          combine_observations)
from .scenario import Scenario


class Action:
    pass


class TeleportAction(Action):
    """action.py:12-63 (note the x, y, z, h, r, p field order of the reference)."""

    def __init__(self, x=0.0, y=0.0, z=0.0, h=0.0, r=0.0, p=0.0, pose: Optional[np.ndarray] = None):
        self.x = pose[0] if pose is not None else x
        self.y = pose[1] if pose is not None else y
        self.z = pose[2] if pose is not None else z
        self.h = pose[3] if pose is not None else h
        self.r = pose[4] if pose is not None else r
        self.p = pose[5] if pose is not None else p

    @property
    def pose(self):
        return np.array([self.x, self.y, self.z, self.h, self.r, self.p])


class VehicleAction(Action):
    """action.py:66-83."""

    def __init__(self, accel: float, steer: float):
        self.acceleration = accel
        self.steering = steer


class Sensor:
    """sensor/base.py:9-54: reset(state) / step(state) -> observation around the _reset / _step hooks."""

    def __init__(self, entity: Entity):
        self.entity = entity

    def reset(self, state):
        return self._reset(state)

    def step(self, state):
        return self._step(state)

    def _reset(self, state):
        return self._step(state)

    def _step(self, state):
        return None


class CombinedSensor(Sensor):
    """sensor/common.py:18-36: several sensors of one entity as one; the observation class is made at reset from the
    classes of the sensors' first observations (combine_observations)."""

    def __init__(self, entity: Entity, *sensors: Sensor):
        assert all(s.entity == entity for s in sensors)
        super().__init__(entity)
        self.sensors = sensors
        self.obs_class = None

    def _reset(self, state):
        first = [s.reset(state) for s in self.sensors]
        self.obs_class = combine_observations(*(type(o) for o in first))
        return self.obs_class.from_obs(*first)

    def _step(self, state):
        return self.obs_class.from_obs(*(s.step(state) for s in self.sensors))


class EgoLocalizationSensor(Sensor):
    """sensor/common.py:39-50: the observation is State.get_entity_data(entity)."""

    def _step(self, state):
        return SingleEntityObservation(self.entity, *state.get_entity_data(self.entity))


class FutureCollisionDetector(Sensor):
    """sensor/common.py:59-106: the localisation observation plus `future_collision` -- does the entity's box, moved along
    its trajectory over `horizon` seconds (10 samples), meet another entity's box at that entity's trajectory position?
    The look-ahead runs on the device for the whole batch (sg_future_collision); the sensor's entity must be the
    scenario's ego."""

    def __init__(self, entity: Entity, horizon: float = 5.0):
        super().__init__(entity)
        self.horizon = horizon

    def _step(self, state):
        if state.scenario.ego is not self.entity:
            raise NotImplementedError("the device look-ahead is evaluated for the ego of each scenario")
        return FutureCollisionObservation(self.entity, *state.get_entity_data(self.entity), state.future_collision(self.horizon))


class RasterizedMapSensor(Sensor):
    """sensor/map.py:26-271: an n x n grid in the entity's frame with one plane per layer -- the bounding boxes of the
    present entities ("entity") and the unions of road-network polygons ("driveable_surface", "road", "intersection",
    "lane", "walkable_surface", "pavement", "crossing") -- computed on the device for the ego of every scenario
    (sg_raster_map).  Default layers as in the reference: entity + driveable_surface."""

    _all_layers = ["entity", "driveable_surface", "road", "intersection", "lane", "walkable_surface", "pavement", "crossing"]

    def __init__(self, entity: Entity, layers=None, height: float = 20.0, width: float = 20.0, freq: Optional[float] = 1.0,
                 n: Optional[int] = None, channels_first: bool = False):
        super().__init__(entity)
        self.layers = ["entity", "driveable_surface"] if layers is None else list(layers)
        for layer in self.layers:
            if layer not in self._all_layers:
                raise NotImplementedError(f"Layer {layer} does not have a get and/or a prepare method.")
        self.height, self.width, self.channels_first = height, width, channels_first
        if n is None:
            assert freq is not None, "At least one of n and freq must be provided."
            self.nw, self.nh = int(freq * width), int(freq * height)
        else:
            self.nw = self.nh = n

    @property
    def output_shape(self):
        return (len(self.layers), self.nw, self.nh) if self.channels_first else (self.nw, self.nh, len(self.layers))

    def _step(self, state):
        if state.scenario.ego is not self.entity:
            raise NotImplementedError("the device raster is evaluated for the ego of each scenario")
        m = state.raster_map(self.layers, self.width, self.height, self.nw, self.nh)
        return MapObservation(self.entity, *state.get_entity_data(self.entity), m if self.channels_first else m.transpose(1, 2, 0))


class GlobalCollisionDetector(Sensor):
    """sensor/common.py:115-129: the localisation observation plus State.collisions() (the device's adjacency rows)."""

    def _step(self, state):
        return CollisionObservation(self.entity, *state.get_entity_data(self.entity), state.collisions())


class Controller:
    """controller.py:12-42: reset(state) / step(state, action) -> pose around the _reset / _step hooks."""

    device_kind = None

    def __init__(self, entity: Entity):
        self.entity = entity

    def reset(self, state) -> None:
        self._reset(state)

    def step(self, state, action):
        return self._step(state, action)

    def _reset(self, state) -> None:
        pass

    def _step(self, state, action):
        raise NotImplementedError

    def ctrl_row(self) -> np.ndarray:
        from .engine import DEFAULT_CTRL

        return DEFAULT_CTRL.copy()


class ReplayTrajectoryController(Controller):
    """controller.py:45-54: pose = action.pose."""

    device_kind = L.KIND_AGENT_REPLAY

    def _step(self, state, action):
        return action.pose


class VehicleController(Controller):
    """controller.py:57-140; driven by external (accel, steer) actions."""

    device_kind = L.KIND_AGENT_VEHICLE

    def __init__(self, entity, max_steer: float = 0.7, max_accel: float = 5.0, max_speed: Optional[float] = None,
                 allow_reverse: bool = False):
        super().__init__(entity)
        self.max_steer, self.max_accel = max_steer, max_accel
        self.max_speed, self.allow_reverse = max_speed, allow_reverse

    def ctrl_row(self):
        row = super().ctrl_row()
        row[L.C_MAX_STEER], row[L.C_MAX_ACCEL] = self.max_steer, self.max_accel
        row[L.C_MAX_SPEED] = np.nan if self.max_speed is None else self.max_speed
        row[L.C_ALLOW_REVERSE] = float(bool(self.allow_reverse))
        return row


class PIDController(VehicleController):
    """controller.py:143-258."""

    device_kind = L.KIND_AGENT_PID

    def __init__(self, entity, steer_Kp: float = 0.03054, steer_Kd: float = 1.5709, accel_Kp: float = 0.3753,
                 accel_Kd: float = 1.8970, accel_Ki: float = 0.0204, **kwargs):
        super().__init__(entity, **kwargs)
        self.steer_Kp, self.steer_Kd = steer_Kp, steer_Kd
        self.accel_Kp, self.accel_Kd, self.accel_Ki = accel_Kp, accel_Kd, accel_Ki

    def ctrl_row(self):
        row = super().ctrl_row()
        row[L.C_STEER_KP], row[L.C_STEER_KD] = self.steer_Kp, self.steer_Kd
        row[L.C_ACCEL_KP], row[L.C_ACCEL_KD], row[L.C_ACCEL_KI] = self.accel_Kp, self.accel_Kd, self.accel_Ki
        return row


class Agent:
    """agent.py:18-116.  The built-in subclasses below are lowered to device lanes; any other subclass runs its
    sensor -> _step -> controller chain on the host every tick (device_kind() == KIND_AGENT_EXTERNAL)."""

    def __init__(self, entity: Entity, controller: Controller, sensor: Sensor):
        self.entity, self.controller, self.sensor = entity, controller, sensor
        self.last_action = None
        self.last_reward = None

    def device_kind(self) -> int:
        """A Python `_step` over the built-in VehicleController: the policy runs in the caller, the controller on the
        device (the (accel, steer) action goes down with every tick, like ExternalVehicleAgent).  Anything else with a
        Python `_step`: the whole chain runs in the caller and the pose is injected (KIND_AGENT_EXTERNAL)."""
        if type(self.controller) is VehicleController and type(self).step is Agent.step:
            return L.KIND_AGENT_VEHICLE
        return L.KIND_AGENT_EXTERNAL

    def host_action(self, state):
        """sensor -> _step for agents whose controller runs on the device: the VehicleAction of this tick."""
        action = self._step(self.sensor.step(state))
        self.last_action = action
        return action

    def reset(self, state) -> None:
        """agent.py:43-50."""
        self.last_action = None
        self.last_reward = None
        self.sensor.reset(state)
        if self.device_kind() == L.KIND_AGENT_EXTERNAL:
            self.controller.reset(state)  # built-in controllers are reset on the device (sg_reset)
        self._reset()

    def step(self, state):
        """agent.py:52-57: observation -> action -> pose."""
        obs = self.sensor.step(state)
        action = self._step(obs)
        self.last_action = action
        return self.controller.step(state, action)

    def _reset(self) -> None:
        pass

    def _step(self, observation):
        raise NotImplementedError

    def reward(self, state):
        r = self._reward(state)
        if r is not None:
            self.last_reward = r
        return r

    def _reward(self, state):
        return None

    def finish(self, state) -> None:
        pass


class ReplayTrajectoryAgent(Agent):
    """agent.py:118-128."""

    def __init__(self, entity, controller=None, sensor=None):
        super().__init__(entity, controller or ReplayTrajectoryController(entity), sensor or EgoLocalizationSensor(entity))

    def device_kind(self):
        return L.KIND_AGENT_REPLAY


class PIDAgent(Agent):
    """agent.py:131-148."""

    def __init__(self, entity, **controller_kwargs):
        super().__init__(entity, PIDController(entity, **controller_kwargs), EgoLocalizationSensor(entity))

    def device_kind(self):
        return L.KIND_AGENT_PID


class ExternalVehicleAgent(Agent):
    """VehicleController driven by caller-supplied (accel, steer) actions: the external-action loop of
    integrations/openaigym.py:171-226 (`ScenarioGym.step(actions)` / `BatchedScenarioGym.step(actions)`)."""

    def __init__(self, entity, **controller_kwargs):
        super().__init__(entity, VehicleController(entity, **controller_kwargs), EgoLocalizationSensor(entity))

    def device_kind(self):
        return L.KIND_AGENT_VEHICLE

    host_action = None  # the caller of step(actions) supplies the action


def _create_agent(scenario: Scenario, entity: Entity) -> Optional[Agent]:
    """Default create_agent (agent.py:151-169): only the entity with ref "ego" gets an agent."""
    if entity.ref == "ego":
        return ReplayTrajectoryAgent(entity)
    return None


# --------------------------------------------------------------------------------------- pedestrians
class BehaviourParameters:
    """pedestrian/behaviour.py:8-15."""

    max_speed_factor = 1.3

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)


class SocialForceParameters(BehaviourParameters):
    """pedestrian/social_force.py:16-30 + random_walk.py:13-19.

    The Gaussian fluctuations (std_lon / std_lat; the reference draws them from numpy's global generator,
    social_force.py:106-108) run on the device.  Two extra attributes choose where the variates come from:
      noise = "device" (default)  a counter-based generator on the GPU keyed by (noise_seed, scenario, entity, step):
                                  the same distribution, its own stream;
      noise = "numpy"             parity with a reference run: scenario i of the batch uses the variates numpy's legacy
                                  generator hands out after np.random.seed(noise_seed + i)
                                  (noise_seed may also be a sequence, one seed per scenario)."""

    bias_lon = 0.0
    bias_lat = 0.0
    std_lon = 0.000002
    std_lat = 0.0000001
    distance_threshold = 3
    sight_weight = 0.5
    sight_weight_use = True
    sight_angle = 200
    relaxation_time = 1.5
    ped_repulse_V = 1.0
    ped_repulse_sigma = 1.0
    ped_attract_C = 0.0
    boundary_repulse_U = 10.0
    boundary_repulse_R = 0.2
    imp_boundary_repulse_U = 2.0
    imp_boundary_repulse_R = 0.1
    noise = "device"
    noise_seed = 0


def _noise_params(p) -> dict:
    noisy = p.std_lon != 0 or p.std_lat != 0
    noise = getattr(p, "noise", "device")
    if noisy and noise not in ("device", "numpy"):
        raise ValueError(f"{type(p).__name__}.noise = {noise!r}: 'device' or 'numpy'")
    return dict(bias_lon=p.bias_lon, bias_lat=p.bias_lat, std_lon=p.std_lon, std_lat=p.std_lat,
                noise=("off" if not noisy else ("stream" if noise == "numpy" else "device")), noise_seed=getattr(p, "noise_seed", 0))


class PedestrianBehaviour:
    """pedestrian/behaviour.py:18-53: the behaviour object only carries its parameters here -- the models run on the device."""

    def __init__(self, params: BehaviourParameters):
        self.params = params
        self.max_speed_factor = params.max_speed_factor

    def device_params(self) -> dict:
        raise NotImplementedError("only SocialForce and RandomWalk are lowered to the device")


class SocialForce(PedestrianBehaviour):
    """pedestrian/social_force.py:33-42."""

    def device_params(self) -> dict:
        p = self.params
        return dict(relaxation_time=p.relaxation_time, ped_repulse_V=p.ped_repulse_V,
                    ped_repulse_sigma=p.ped_repulse_sigma, ped_attract_C=p.ped_attract_C,
                    sight_weight=p.sight_weight, sight_weight_use=p.sight_weight_use, sight_angle=p.sight_angle,
                    max_speed_factor=p.max_speed_factor,
                    imp_boundary_repulse_U=p.imp_boundary_repulse_U, imp_boundary_repulse_R=p.imp_boundary_repulse_R,
                    behaviour="social_force", **_noise_params(p))


class RandomWalkParameters(BehaviourParameters):
    """pedestrian/random_walk.py:13-19 (+ `noise` / `noise_seed`: where the Gaussian variates come from, as in
    SocialForceParameters)."""

    bias_lon = 0.0
    bias_lat = 0.0
    std_lon = 0.000002
    std_lat = 0.0000001
    noise = "device"
    noise_seed = 0


class RandomWalk(PedestrianBehaviour):
    """pedestrian/random_walk.py:22-44: speed ~ N(speed_desired + bias_lon, std_lon), heading ~ N(angle to the goal point +
    bias_lat, std_lat); no neighbours, no max_speed_factor (the controller's max_speed alone clips)."""

    def __init__(self, params: RandomWalkParameters):
        super().__init__(params)
        self.bias_lon, self.bias_lat, self.std_lon, self.std_lat = params.bias_lon, params.bias_lat, params.std_lon, params.std_lat

    def device_params(self) -> dict:
        return dict(max_speed_factor=self.params.max_speed_factor, behaviour="random_walk", **_noise_params(self.params))


class PedestrianSensor(Sensor):
    """pedestrian/sensor.py:12-40."""

    def __init__(self, entity, head_rot_angle: float = 0.0, distance_threshold: float = 1.0):
        super().__init__(entity)
        self.head_rot_angle = head_rot_angle
        self.distance_threshold = distance_threshold


class PedestrianController(Controller):
    """pedestrian/controller.py:9-46."""

    device_kind = L.KIND_AGENT_PEDESTRIAN

    def __init__(self, entity, max_speed: float = 5.0):
        super().__init__(entity)
        self.max_speed = max_speed


class PedestrianAgent(Agent):
    """pedestrian/agent.py:15-69: follows `route` with a behaviour model (SocialForce or RandomWalk)."""

    def __init__(self, entity, route, speed_desired: float, behaviour: PedestrianBehaviour, max_speed: float = 5.0,
                 head_rot_angle: float = 0.0, distance_threshold: float = 1.0):
        super().__init__(entity, PedestrianController(entity, max_speed=max_speed),
                         PedestrianSensor(entity, head_rot_angle=head_rot_angle, distance_threshold=distance_threshold))
        if not isinstance(behaviour, (SocialForce, RandomWalk)):
            raise NotImplementedError("only the SocialForce and RandomWalk behaviours are lowered to the device")
        self.speed_desired = speed_desired
        self.behaviour = behaviour
        self.route = np.asarray(route, np.float64).reshape(-1, 2)
        self._bound = None  # (gym, scenario index, entity slot) once the scenario is on the device

    def _bind(self, gym, i: int, slot: int):
        self._bound = (gym, i, slot)

    @property
    def force(self) -> np.ndarray:
        """PedestrianAgent.force (pedestrian/agent.py:63): the social force of the latest step, read from the device state."""
        if self._bound is None:
            return np.array([0.0, 0.0])
        gym, i, slot = self._bound
        return gym._fetch_state()["force"][i, slot].copy()

    @property
    def speed(self) -> float:
        """PedestrianController.speed (pedestrian/controller.py:39): the clipped speed of the latest step, from the device state."""
        if self._bound is None:
            return 0.0
        gym, i, slot = self._bound
        return float(gym._fetch_state()["ctrl_state"][i, slot, 0])

    @property
    def goal_idx(self) -> int:
        """Index of the waypoint the pedestrian walks towards (pedestrian/agent.py:59-62), from the device state."""
        if self._bound is None:
            return 0
        gym, i, slot = self._bound
        return int(gym._fetch_state()["ctrl_state"][i, slot, 1])

    def device_kind(self):
        return L.KIND_AGENT_PEDESTRIAN

    def ctrl_row(self):
        row = self.controller.ctrl_row()
        row[L.C_PED_SPEED_DESIRED] = self.speed_desired
        row[L.C_PED_MAX_SPEED] = self.controller.max_speed
        row[L.C_PED_HEAD_ROT] = self.sensor.head_rot_angle
        row[L.C_PED_RADIUS] = self.sensor.distance_threshold
        return row
