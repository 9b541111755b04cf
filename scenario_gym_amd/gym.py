"""ScenarioGym / BatchedScenarioGym: the reference's orchestrator API over the device engine.

`ScenarioGym` keeps the reference's constructor and methods (scenario_gym/scenario_gym.py:13-319) for ONE
scenario; `BatchedScenarioGym` is the batched front door (a list of scenarios = one device batch) that
replaces the reference's `for scenario in scenarios: gym.rollout()` loops (scenario_gym.py:16-27,
manager.py:272-282).  With only built-in agents / metrics / terminal conditions a rollout is ONE
kernel launch; user-defined Python metrics, state callbacks or callable terminal conditions switch to
one launch per step with the State view refreshed in between.
"""
import math
from typing import Any, Callable, Dict, List, Optional, Sequence, Union

import numpy as np

from .agent import _create_agent
from .engine import TERMINAL_BITS, RolloutEngine
from .metrics import RSS, CollisionPointMetric, Metric, RSSDistances, _DeviceMetric
from .packing import pack_scenarios
from .scenario import Scenario
from .state import State


class BatchedScenarioGym:
    def __init__(self, timestep: float = 1.0 / 30.0, persist: bool = False,
                 terminal_conditions: Optional[List[Union[str, Callable]]] = None,
                 state_callbacks: Optional[List[Callable]] = None,
                 metrics: Optional[Callable[[], List[Metric]]] = None,
                 record: bool = False, event_capacity: int = 16, device: int = 0):
        """`metrics` is a factory returning a fresh list of Metric objects (one list per scenario)."""
        self._timestep = float(timestep)
        self.persist = persist
        self.terminal_conditions = ["max_length"] if terminal_conditions is None else list(terminal_conditions)
        self.state_callbacks = state_callbacks or []
        # RSSDistances runs inside the library after every step (sg_set_rss); every other callback is a host callable
        self._host_callbacks = [cb for cb in self.state_callbacks if not isinstance(cb, RSSDistances)]
        self._rss_on = len(self._host_callbacks) != len(self.state_callbacks)
        self.metric_factory = metrics or (lambda: [])
        self.record = record
        self.event_capacity = event_capacity
        self.device = device
        self.engine: Optional[RolloutEngine] = None
        self.scenarios: List[Scenario] = []
        self._host_agents: list = []
        self._policy_agents: list = []
        self.states: List[State] = []
        self.metrics: List[List[Metric]] = []
        self._cache = None
        self._prev_state = None
        self._rec = None
        self._fut = None
        self._rss_cache = None

    # ------------------------------------------------------------------ properties
    @property
    def timestep(self) -> float:
        return self._timestep

    @timestep.setter
    def timestep(self, dt: float) -> None:
        self._timestep = float(dt)
        if self.engine is not None:
            self.engine.set_timestep(dt)

    def _host_terminals(self):
        return [c for c in self.terminal_conditions if callable(c)]

    def _per_step_host_path(self) -> bool:
        custom_metric = any(not isinstance(m, _DeviceMetric) for ms in self.metrics for m in ms)
        return bool(custom_metric or self._host_callbacks or self._host_terminals() or self._host_agents
                    or self._policy_agents or self._custom_actions())

    def _custom_actions(self) -> bool:
        """Some scenario carries an action whose trigger / effect is the caller's own code (not one of the time-triggered
        classes of actions.py): State.update_actions then runs after every tick, like any other Python extension."""
        from .actions import time_triggered

        return any(time_triggered(a) is None for st in getattr(self, "states", ()) for a in st.scenario.actions)

    def _replay_actions(self, before):
        """State.update_actions for the steps a multi-step device call just ran: the clock of scenario i after each of them is
        t + dt + dt + ... (scenario_gym.py:229, the additions of the step loop), so the time-triggered actions fire at the
        same State.t -- bit for bit -- as when the scenario is stepped one tick at a time."""
        if not any(st.unapplied_actions for st in self.states):
            return
        now = self._fetch_state()
        for i, st in enumerate(self.states):
            if not st.unapplied_actions:
                continue
            t, k = float(before["t"][i]), int(now["n_steps"][i]) - int(before["n_steps"][i])
            clock = []
            for _ in range(max(k, 0)):
                t = t + self._timestep
                clock.append(t)
            st._replay_actions(clock)

    def _push_host_agents(self, actions):
        """scenario_gym.py:233-239 for the agents that run in Python.  Pose agents: agent.step(state) of every present
        one, its pose (or None -> NaN) handed to the next device step.  Policy agents (Python `_step` over the built-in
        VehicleController): their (accel, steer) of this tick; returns the action array for sg_step."""
        if self._host_agents:
            poses = np.full((len(self.states), self._packed.n_entities, 6), np.nan)
            for i, slot, agent in self._host_agents:
                st = self.states[i]
                if agent.entity in st.poses:
                    pose = agent.step(st)
                    if pose is not None:
                        poses[i, slot] = np.asarray(pose, np.float64)
            self.engine.set_external_poses(poses)
        if self._policy_agents:
            actions = np.zeros((1, len(self.states), 2)) if actions is None else np.array(actions, np.float64).reshape(1, -1, 2)
            for i, agent in self._policy_agents:
                st = self.states[i]
                if agent.entity in st.poses:
                    a = agent.host_action(st)
                    actions[0, i] = (a.acceleration, a.steering)
        return actions

    # ------------------------------------------------------------------ set up
    @classmethod
    def run_scenarios(cls, paths: Sequence[str], create_agent=_create_agent, relabel: bool = True, workers: int = 8,
                      **kwargs) -> List[Dict[str, Any]]:
        """ScenarioGym.run_scenarios (scenario_gym.py:16-27) for a list of OpenSCENARIO files as ONE device batch:
        parse (in parallel), pack, roll every scenario out, return the metrics of each."""
        gym = cls(**kwargs)
        try:
            gym.load_scenarios(paths, create_agent=create_agent, relabel=relabel, workers=workers)
            gym.rollout()
            return gym.get_metrics()
        finally:
            gym.close()

    def load_scenarios(self, paths: Sequence[str], create_agent=_create_agent, relabel: bool = True, workers: Optional[int] = None,
                       max_steps: Optional[int] = None, processes: bool = False):
        """ScenarioGym.load_scenario (scenario_gym.py:119-155) for many files (.xosc, or .json written by Scenario.to_json):
        every OpenSCENARIO file goes through the native scan
        (libsgym_xosc.so, which runs without the GIL) on `workers` threads -- or, for directories of thousands of files,
        `workers` processes (`processes=True`: the Python side of the import scales too; tools/ingest_rate.py) -- and the
        batch is packed once.  workers=None: 8 threads, or as many processes as the CPU quota grants."""
        from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor
        from functools import partial

        from .xosc import load_scenario_file

        paths = list(paths)
        if workers is None:
            from .packing import effective_cpus

            workers = effective_cpus() if processes else 8
        load = partial(load_scenario_file, relabel=relabel)
        if workers > 1 and len(paths) > 1 and processes:
            # "spawn": a forked child would inherit this process's HIP runtime (its threads' locks, pinned buffers) once
            # an engine exists; the importer is importable on its own, so fresh interpreters work
            import multiprocessing

            with ProcessPoolExecutor(min(workers, len(paths)), mp_context=multiprocessing.get_context("spawn")) as ex:
                scenarios = list(ex.map(load, paths, chunksize=32))
        elif workers > 1 and len(paths) > 1:
            with ThreadPoolExecutor(min(workers, len(paths))) as ex:
                scenarios = list(ex.map(load, paths))
        else:
            scenarios = [load(f) for f in paths]
        self.set_scenarios(scenarios, create_agent=create_agent, max_steps=max_steps)
        for st, f in zip(self.states, paths):
            st.scenario_path = f

    def set_packed(self, packed, max_steps: Optional[int] = None):
        """Throughput path (ScenarioManager.run_scenarios-style sweeps, manager.py:240-283): a batch that arrives already
        packed -- `packing.load_and_pack` in worker processes, `packing.merge_packed` here -- with the reference's default
        agents (ego replay agent, everyone else batch replay).  No Scenario / State objects are built: rollout() and
        get_metrics() (device metrics only) are what such a sweep calls.  The engine of the previous batch is reused when the
        shape is the same."""
        if packed.kind.max(initial=0) > 2:
            raise ValueError("set_packed: replay entities and replay agents only (the default create_agent)")
        if self._host_callbacks or self._host_terminals():
            raise ValueError("set_packed: device callbacks / terminal conditions only")
        horizon = float(np.max(packed.length - packed.t0))
        self.max_steps = int(max_steps or (math.ceil(max(horizon, 0.0) / self._timestep) + 8))
        dev_terms = list(self.terminal_conditions)
        reuse = (self.engine is not None and not self.record and not self.states and
                 (self.engine.R, self.engine.E) == (packed.n_scenarios, packed.n_entities))
        if not reuse:
            self.close()
            self.engine = RolloutEngine(packed.n_scenarios, packed.n_entities, timestep=self._timestep, persist=self.persist,
                                        terminal_conditions=dev_terms, record_capacity=0, event_capacity=self.event_capacity,
                                        device=self.device)
            if self._rss_on:
                self.engine.set_rss(True)
        self.engine.upload(packed)
        self._packed = packed
        self.scenarios, self.states, self._host_agents, self._policy_agents = [], [], [], []
        self._roads_set = True
        self.metrics = [list(self.metric_factory()) for _ in range(packed.n_scenarios)]
        if any(not isinstance(m, _DeviceMetric) for m in self.metrics[0]):
            raise ValueError("set_packed: device metrics only")
        self._invalidate()
        self._prev_state = None

    def set_scenarios(self, scenarios: Sequence[Scenario], create_agent=_create_agent, max_steps: Optional[int] = None):
        self.close()
        self.scenarios = list(scenarios)
        import time as _time

        _t = _time.perf_counter()
        packed, agents = pack_scenarios(self.scenarios, create_agent)
        self._timings = {"pack": _time.perf_counter() - _t}
        self._packed = packed
        horizon = float(np.max(packed.length - packed.t0))
        self.max_steps = int(max_steps or (math.ceil(max(horizon, 0.0) / self._timestep) + 8))
        dev_terms = [c for c in self.terminal_conditions if not callable(c)]
        for c in dev_terms:
            if c not in TERMINAL_BITS:
                raise ValueError(f"terminal condition {c!r} is not supported")
        # behaviour models: every PedestrianAgent holds its own behaviour object (pedestrian/agent.py:18-41).  The distinct
        # (behaviour, parameters, std) combinations of the batch become the device's models (sg_set_ped_models), every
        # pedestrian slot carries the index of its own; one combination: the handle-wide parameter set, as before
        sf = None
        models, keys, model_of = [], {}, np.zeros(packed.n_scenarios * packed.n_entities, np.int32)
        for i, (sc, sc_agents) in enumerate(zip(self.scenarios, agents)):
            for e, a in sc_agents.items():
                if hasattr(a, "behaviour"):
                    dp = a.behaviour.device_params()
                    if sf is None:
                        sf = dict(dp)
                    elif (dp["noise"], dp["noise_seed"] if not hasattr(dp["noise_seed"], "__len__") else 0) != \
                            (sf["noise"], sf["noise_seed"] if not hasattr(sf["noise_seed"], "__len__") else 0) and \
                            "off" not in (dp["noise"], sf["noise"]):
                        raise NotImplementedError("pedestrian behaviours with different noise sources (noise / noise_seed) in one gym: the "
                                                  "reference draws every agent's variates from the one global generator")
                    if sf["noise"] == "off" and dp["noise"] != "off":  # (a std-0 model beside a noisy one: the noisy one names the source)
                        sf["noise"], sf["noise_seed"] = dp["noise"], dp["noise_seed"]
                    key = tuple(sorted((k, v) for k, v in dp.items() if k not in ("noise", "noise_seed")))
                    if key not in keys:
                        keys[key] = len(models)
                        models.append({k: v for k, v in dp.items() if k not in ("noise", "noise_seed")})
                    model_of[i * packed.n_entities + sc.entities.index(e)] = keys[key]
        self._n_ped_models = len(models)
        if len(models) > 1:
            sf = dict(models=models, model_of=model_of, noise=sf["noise"], noise_seed=sf["noise_seed"])
        if sf is not None and sf.get("noise") == "stream":
            # parity with the reference's global generator: scenario i draws what numpy hands out after seed(noise_seed + i)
            seeds = sf.pop("noise_seed")
            seeds = list(seeds) if hasattr(seeds, "__len__") else [int(seeds) + i for i in range(packed.n_scenarios)]
            n_ped = int((packed.kind.reshape(packed.n_scenarios, -1) == 5).sum(axis=1).max())
            n = 2 * n_ped * (self.max_steps + 1)
            sf["normals"] = np.stack([np.random.RandomState(int(k)).standard_normal(n) for k in seeds])
        self.engine = RolloutEngine(
            packed.n_scenarios, packed.n_entities, timestep=self._timestep, persist=self.persist,
            terminal_conditions=dev_terms, record_capacity=(self.max_steps + 1) if self.record else 0,
            event_capacity=self.event_capacity, device=self.device, social_force=sf)
        if self._rss_on:
            self.engine.set_rss(True)
        self.engine.upload(packed)
        self._roads_set = False
        # the road network reaches the device when the rollout itself needs it: the ego_off_road terminal condition, and
        # pedestrian agents (the boundary terms of the social force, social_force.py:86-104)
        if "ego_off_road" in dev_terms or (sf is not None and any(sc.road_network is not None for sc in self.scenarios)):
            self._set_road_networks()
        self.states = [State(self, i, sc, agents[i], self.persist) for i, sc in enumerate(self.scenarios)]
        for i, sc in enumerate(self.scenarios):
            for e, a in agents[i].items():
                if hasattr(a, "_bind"):
                    a._bind(self, i, sc.entities.index(e))
        from . import _lib as L
        self._host_agents = [(i, sc.entities.index(e), a) for i, sc in enumerate(self.scenarios)
                             for e, a in agents[i].items() if a.device_kind() == L.KIND_AGENT_EXTERNAL]
        # a Python policy over the device VehicleController: the action array carries one (accel, steer) per scenario
        self._policy_agents = [(i, a) for i, sc in enumerate(self.scenarios) for e, a in agents[i].items()
                               if a.device_kind() == L.KIND_AGENT_VEHICLE and getattr(a, "host_action", None) is not None]
        if len({i for i, _ in self._policy_agents}) != len(self._policy_agents):
            raise NotImplementedError("one external-action vehicle agent per scenario (sg_step actions are [n][R][2])")
        self.metrics = [list(self.metric_factory()) for _ in self.scenarios]
        self._invalidate()
        self._prev_state = None
        self._reset_host_side()

    def engine_models(self) -> int:
        """Distinct pedestrian behaviour models (behaviour, parameters, std) of the loaded batch (0: no pedestrian agents)."""
        return getattr(self, "_n_ped_models", 0)

    def _set_road_networks(self):
        """Scenario.road_network of every scenario -> the device (shared networks once); needed by the ego_off_road
        terminal condition and the surface layers of the map sensor."""
        if self._roads_set:
            return
        nets, index, net_of = [], {}, []
        for sc in self.scenarios:
            rn = sc.road_network
            if rn is None:
                net_of.append(-1)
                continue
            if id(rn) not in index:
                index[id(rn)] = len(nets)
                nets.append(rn.polygon_arrays())
            net_of.append(index[id(rn)])
        self.engine.set_road_networks(nets, net_of)
        self._roads_set = True
        # flags cached before the upload were computed without a road index (ego_off_road set for everybody): the answer must
        # not depend on which terminal condition was asked first
        if self._fut is not None:
            self._fut.pop(("term",), None)

    def _raster_map(self, layers, width, height, nw, nh):
        """[R][n_layers][nh][nw] of RasterizedMapSensor layers (names of sensor/map.py:44-53), cached per state."""
        from .road_network import LAYER_CODES
        codes = tuple(LAYER_CODES[l] for l in layers)
        key = ("map", codes, width, height, nw, nh)
        if self._fut is None or key not in self._fut:
            if any(codes):
                self._set_road_networks()
            self._fut = dict(self._fut or {})
            self._fut[key] = self.engine.raster_map(codes, width, height, nw, nh)
        return self._fut[key]

    def _rss_results(self):
        if self._rss_cache is None:
            self._rss_cache = self.engine.rss()
        return self._rss_cache

    def _invalidate(self):
        self._rss_cache = None
        self._cache = None
        self._rec = None
        self._fut = None

    def _fetch_state(self):
        if self._cache is None:
            self._cache = self.engine.state()
        return self._cache

    def _terminal_flags(self):
        """SG_TERM_* bits of every scenario's current state (all four conditions), cached per state."""
        key = ("term",)
        if self._fut is None or key not in self._fut:
            self._fut = dict(self._fut or {})
            self._fut[key] = self.engine.terminal_flags()
        return self._fut[key]

    def _future(self, horizon: float, n_samples: int):
        key = (horizon, n_samples)
        if self._fut is None or key not in self._fut:
            self._fut = dict(self._fut or {})
            self._fut[key] = self.engine.future_collision(horizon, n_samples)
        return self._fut[key]

    def _raster(self, width, height, nw, nh):
        key = ("raster", width, height, nw, nh)
        if self._fut is None or key not in self._fut:
            self._fut = dict(self._fut or {})
            self._fut[key] = self.engine.raster_entities(width, height, nw, nh)
        return self._fut[key]

    def _fetch_record(self):
        if not self.record:
            raise RuntimeError("recorded_poses needs BatchedScenarioGym(record=True)")
        if self._rec is None:
            n = int(self._fetch_state()["n_steps"].max()) + 1
            self._rec = self.engine.record(min(n, self.max_steps + 1))
        return self._rec

    def _reset_host_side(self):
        for st in self.states:  # State.reset: _reset_data, ..., update_actions() at t0 (state.py:106-143)
            st._reset_actions()
            if st.unapplied_actions:
                st.update_actions()
        for i, _, agent in self._host_agents:  # Agent.reset -> sensor / controller reset (scenario_gym.py:217-225)
            agent.reset(self.states[i])
        for i, agent in self._policy_agents:
            agent.reset(self.states[i])
        for st, ms in zip(self.states, self.metrics):
            for m in ms:
                m.reset(st)
            for cb in self._host_callbacks:
                if hasattr(cb, "reset"):
                    cb.reset(st)
                cb(st)

    def reset_scenarios(self):
        self.engine.reset()
        self._invalidate()
        self._prev_state = None
        self._reset_host_side()

    # ------------------------------------------------------------------ stepping
    def _after_host_step(self):
        done_host = np.zeros(len(self.states), bool)
        for i, (st, ms) in enumerate(zip(self.states, self.metrics)):
            if st.unapplied_actions:  # State.step: update_poses, update_actions, update_callbacks, check_terminal (state.py:165-171)
                st.update_actions()
            for cb in self._host_callbacks:
                cb(st)
            done_host[i] = any(c(st) for c in self._host_terminals())
            for m in ms:
                if not isinstance(m, _DeviceMetric):
                    m.step(st)
        return done_host

    def step(self, actions=None, n: int = 1):
        """n x ScenarioGym.step() for every scenario; actions [n, R, 2] for ExternalVehicleAgent egos."""
        if n == 1 or not self._per_step_host_path():
            has_actions = any(st.unapplied_actions for st in self.states)
            self._prev_state = self._fetch_state() if self._per_step_host_path() or n == 1 or has_actions else None
            before = self._prev_state
            if n == 1:
                actions = self._push_host_agents(actions)
            self.engine.step(n, actions)
            self._invalidate()
            if self._per_step_host_path():
                self._after_host_step()
            elif has_actions:
                self._replay_actions(before)
            if n > 1 and not self._per_step_host_path():
                self._prev_state = None
            return
        actions = None if actions is None else np.asarray(actions, np.float64).reshape(n, len(self.states), 2)
        for k in range(n):
            self.step(None if actions is None else actions[k:k + 1], 1)

    def rollout(self, max_steps: Optional[int] = None):
        """ScenarioGym.rollout() for the whole batch (scenario_gym.py:256-267)."""
        max_steps = int(max_steps or self.max_steps)
        if not self._per_step_host_path():
            self.engine.rollout(max_steps)
            self._invalidate()
            if any(st.scenario.actions for st in self.states):  # the reset's update_actions(), then one per executed step
                now = self._fetch_state()
                for st in self.states:
                    st._reset_actions()
                # (the clock at the reset: what the device holds minus the executed steps is not exact -- take it from t0)
                t0 = getattr(self._packed, "t0", None)
                for i, st in enumerate(self.states):
                    if not st.unapplied_actions:
                        continue
                    t = float(t0[i]) if t0 is not None else max(0.0, st.scenario.ego.trajectory.min_t)
                    clock = [t]
                    for _ in range(int(now["n_steps"][i])):
                        t = t + self._timestep
                        clock.append(t)
                    st._replay_actions(clock)
        else:
            self.reset_scenarios()
            done = np.zeros(len(self.states), bool)
            for _ in range(max_steps):
                # a scenario that finished keeps stepping on the device but is frozen for the caller
                self._prev_state = self._fetch_state()
                self.engine.step(1, self._push_host_agents(None))
                self._invalidate()
                done |= self._after_host_step() | self._fetch_state()["done"]
                if done.all():
                    break
        for st in self.states:
            for agent in st.agents.values():
                agent.finish(st)

    # ------------------------------------------------------------------ results
    def get_metrics(self) -> List[Dict[str, Any]]:
        """ScenarioGym.get_metrics (scenario_gym.py:308-319) per scenario."""
        tols = {m.c_tol for ms in self.metrics for m in ms if hasattr(m, "c_tol")}
        if len(tols) > 1:
            raise NotImplementedError("one CollisionMetric.c_tol per batch")
        if tols:
            self.engine._check(self.engine.lib.sg_set_collision_tolerance(self.engine.h, float(tols.pop())),
                               "sg_set_collision_tolerance")
        rows, events = self.engine.metrics()
        want_points = any(isinstance(m, CollisionPointMetric) for ms in self.metrics for m in ms)
        points = self.engine.collision_points() if want_points else None
        rss = None
        if any(isinstance(m, RSS) for ms in self.metrics for m in ms):
            if not any(isinstance(cb, RSSDistances) for cb in self.state_callbacks):
                raise ValueError("Callback RSSDistances is required for RSS.")  # StateCallback.reset, callback.py:27-33
            rss = self._rss_results()
        out = []
        for i, ms in enumerate(self.metrics):
            sel = events["scenario"] == i
            ev = events[sel]
            values = {}
            for m in ms:
                if isinstance(m, CollisionPointMetric):
                    m._load(rows[i], ev, self._packed.refs[i], points[sel])
                elif isinstance(m, RSS):
                    m._load(rows[i], ev, self._packed.refs[i], (rss[0][i], rss[1][i]))
                elif isinstance(m, _DeviceMetric):
                    m._load(rows[i], ev, self._packed.refs[i])
                v = m.get_state()
                if isinstance(v, dict):
                    values.update({f"{m.name}_{k}": x for k, x in v.items() if isinstance(k, str)})
                elif v is not None:
                    values[m.name] = v
            out.append(values)
        return out

    def metric_rows(self):
        """Raw per-scenario sg_metrics rows + the ego collision event table (numpy structured arrays)."""
        return self.engine.metrics()

    def close(self):
        if self.engine is not None:
            self.engine.close()
            self.engine = None


class ScenarioGym:
    """The reference's ScenarioGym for one scenario at a time (scenario_gym/scenario_gym.py:13-319)."""

    @classmethod
    def run_scenarios(cls, paths: List[str], render: bool = False, **kwargs) -> None:
        gym = cls(**kwargs)
        for path in paths:
            gym.load_scenario(path)
            gym.rollout(render=render)

    def __init__(self, timestep: float = 1.0 / 30.0, persist: bool = False, viewer_class=None,
                 terminal_conditions=None, state_callbacks=None, metrics: Optional[List[Metric]] = None,
                 device: int = 0, **viewer_parameters):
        if viewer_class is not None:
            raise NotImplementedError("rendering is outside the device rollout path")
        self._metrics: List[Metric] = list(metrics or [])
        self._b = BatchedScenarioGym(timestep=timestep, persist=persist, terminal_conditions=terminal_conditions,
                                     state_callbacks=state_callbacks, metrics=lambda: self._metrics,
                                     record=True, event_capacity=256, device=device)
        self.state: Optional[State] = None

    timestep = property(lambda self: self._b.timestep, lambda self, dt: setattr(self._b, "timestep", dt))
    persist = property(lambda self: self._b.persist)
    metrics = property(lambda self: self._metrics)
    terminal_conditions = property(lambda self: self._b.terminal_conditions)

    def add_metrics(self, metrics: List[Metric]) -> None:
        self._metrics.extend(metrics)

    def reset_gym(self) -> None:
        self._b.close()
        self.state = None
        self._metrics.clear()

    def load_scenario(self, scenario_path: str, create_agent=_create_agent, relabel: bool = False, **kwargs) -> None:
        from .xosc import load_scenario_file

        self.set_scenario(load_scenario_file(scenario_path, relabel=relabel, **kwargs), scenario_path, create_agent)

    def set_scenario(self, scenario: Scenario, scenario_path: Optional[str] = None, create_agent=_create_agent) -> None:
        self._b.set_scenarios([scenario], create_agent=create_agent)
        self.state = self._b.states[0]
        self.state.scenario_path = scenario_path

    def get_start_time(self, scenario: Scenario) -> float:
        return max((0.0, scenario.ego.trajectory.min_t))

    def reset_scenario(self) -> None:
        if self.state is not None and self.state.t != self.get_start_time(self.state.scenario):
            self._b.reset_scenarios()

    def step(self, action=None) -> None:
        """One tick; `action` = (accel, steer) when the ego is an ExternalVehicleAgent."""
        self._b.step(None if action is None else np.asarray(action, np.float64).reshape(1, 1, 2))

    def rollout(self, render: bool = False, video_path: Optional[str] = None) -> None:
        if render:
            raise NotImplementedError("rendering is outside the device rollout path")
        self._b.rollout()

    def get_metrics(self) -> Dict[str, Any]:
        return self._b.get_metrics()[0]

    def close(self) -> None:
        pass
