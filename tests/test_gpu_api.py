"""GPU tests of the scenario_gym-shaped API (ScenarioGym / BatchedScenarioGym / State / Metric views):
they read like the reference's own tests (tests/test_metrics.py, test_state.py, test_scenario_gym.py,
test_controller.py, test_utils.py) and are checked against the same golden vectors."""
import os

import numpy as np
import pytest

from conftest import ROOT, bits_equal, load_golden, scenario_arrays
from test_host_api import scenario_from_arrays

pytestmark = pytest.mark.gpu


def _scenario(g, prefix):
    return scenario_from_arrays(scenario_arrays(g, prefix), g[prefix + "/refs"])


def test_metrics_on_xosc_scenario():
    """tests/test_metrics.py:13-34 on scenario 3fee6507 with the default gym."""
    import scenario_gym_amd as sga

    g = load_golden("scenarios")
    gym = sga.ScenarioGym(metrics=[sga.EgoAvgSpeed(), sga.EgoMaxSpeed(), sga.EgoDistanceTravelled(), sga.CollisionMetric()])
    gym.set_scenario(_scenario(g, "3fee6507/scenario"))
    gym.rollout()
    m = gym.get_metrics()
    assert 4 <= m["ego_avg_speed"] <= 5 and 10 <= m["ego_max_speed"] <= 12
    assert 90 <= m["ego_distance_travelled"] <= 110 and m["collisions"] == []
    for k in ("ego_avg_speed", "ego_max_speed", "ego_distance_travelled"):
        assert m[k] == float(g[f"3fee6507/dt30/metric_{k}"])  # bit-identical to the reference
    assert gym.state.is_done and gym.state.t == g["3fee6507/dt30/t"][-1]


def test_state_view_matches_reference_bookkeeping():
    """tests/test_state.py:32-73, 103-163: velocities == delta pose / dt, recorded poses, vanishing."""
    import scenario_gym_amd as sga

    g = load_golden("scenarios")
    sc = _scenario(g, "vanish/scenario")
    gym = sga.ScenarioGym(timestep=0.1)
    gym.set_scenario(sc)
    ref = "vanish/nopersist"
    ents = sc.entities
    assert gym.state.t == g[ref + "/t"][0]
    for e in ents:  # tests/test_scenario_gym.py:57-65
        if e.trajectory.min_t <= gym.state.t <= e.trajectory.max_t or e.is_static():
            assert e in gym.state.poses
    for k in range(1, 30):
        gym.step()
        st = gym.state
        assert st.t == g[ref + "/t"][k] and np.isclose(st.dt, 0.1)
        poses, prev, vels = st.poses, st.prev_poses, st.velocities
        for i, e in enumerate(ents):
            gp = g[ref + "/poses"][k, i]
            assert (e in poses) == (not np.isnan(gp[0]))
            if e in poses:
                assert bits_equal(poses[e], gp) and bits_equal(vels[e], g[ref + "/vels"][k, i])
                if e in prev:
                    assert np.array_equal(vels[e], (poses[e] - prev[e]) / st.dt)
            assert st.distances[e] == g[ref + "/dists"][k, i]
    gym.rollout()
    assert ents[1] not in gym.state.poses  # tests/test_scenario_gym.py:68-73
    rec = gym.state.recorded_poses(ents[0])
    n = int(g[ref + "/n_steps"])
    assert rec.shape == (n + 1, 7) and bits_equal(rec[:, 0], g[ref + "/t"]) and bits_equal(rec[:, 1:], g[ref + "/poses"][:, 0])
    all_rec = gym.state.recorded_poses()
    assert len(all_rec[ents[1]]) == int((~np.isnan(g[ref + "/poses"][:, 1, 0])).sum())


def test_persist_keeps_every_entity():
    """tests/test_scenario_gym.py:76-97."""
    import scenario_gym_amd as sga

    g = load_golden("scenarios")
    sc = _scenario(g, "vanish/scenario")
    gym = sga.ScenarioGym(timestep=0.1, persist=True)
    gym.set_scenario(sc)
    assert len(gym.state.poses) == len(sc.entities)
    while not gym.state.is_done:
        gym.step()
        assert len(gym.state.poses) == len(sc.entities)
    assert gym.state.t == g["vanish/persist/t"][-1]


def test_pid_agent_rollout():
    """tests/test_controller.py:7-25."""
    import scenario_gym_amd as sga

    g = load_golden("pid_xosc")
    sc = _scenario(g, "scenario")

    def create_agent(s, e):
        if e.ref == "ego":
            return sga.PIDAgent(e, accel_Kp=2.0, max_accel=5.0, max_steer=np.pi / 90)

    gym = sga.ScenarioGym(timestep=0.1)
    gym.set_scenario(sc, create_agent=create_agent)
    gym.rollout()
    assert np.abs(gym.state.poses[sc.ego] - g["run/poses"][-1, 0]).max() < 1e-9
    assert np.abs(gym.state.recorded_poses(sc.ego)[:, 1:] - g["run/poses"][:, 0]).max() < 1e-9


def test_collision_scene_and_terminal_condition():
    """tests/test_utils.py:12-61 + tests/test_scenario_gym.py:28-31 (terminal_conditions)."""
    import scenario_gym_amd as sga

    g = load_golden("collision")
    sc = _scenario(g, "headon/scenario")
    ego, haz = sc.entities
    gym = sga.ScenarioGym(timestep=0.1, metrics=[sga.CollisionMetric()])
    gym.set_scenario(sc)
    assert not gym.state.collisions()[ego]
    gym.rollout()
    assert gym.state.collisions()[ego] == [haz] and gym.state.collisions()[haz] == [ego]
    assert gym.get_metrics()["collisions"] == [(8.799999999999985, "entity_1", "non_vehicle")]
    obs = sga.GlobalCollisionDetector(ego).step(gym.state)  # sensor/common.py:115-129
    assert obs.collisions == gym.state.collisions() and np.array_equal(obs.pose, gym.state.poses[ego]) and obs.entity is ego
    gym2 = sga.ScenarioGym(timestep=0.1, terminal_conditions=["max_length", "collision"])
    gym2.set_scenario(sc)
    gym2.rollout()
    assert gym2.state.t == 8.799999999999985  # stops at the first contact


def test_timestep_can_change_between_steps():
    """tests/test_scenario_gym.py:28-44."""
    import scenario_gym_amd as sga

    g = load_golden("scenarios")
    gym = sga.ScenarioGym(timestep=0.5)
    gym.set_scenario(_scenario(g, "a5e43fe4/scenario"))
    gym.rollout()
    gym.reset_scenario()
    gym.step()
    assert np.allclose(gym.state.dt, 0.5)
    gym.timestep = 0.2
    gym.step()
    assert np.allclose(gym.state.t, gym.state.prev_t + 0.2)


def test_user_defined_python_metric_and_callback():
    """The Metric / StateCallback extension API (metrics/base.py:8-73, callback.py:9-41): a Python
    metric forces one launch per step and sees the same State the reference would show it."""
    import scenario_gym_amd as sga

    class MaxOthersSpeed(sga.Metric):
        name = "max_other_speed"

        def _reset(self, state):
            self.v, self.steps = 0.0, 0

        def _step(self, state):
            self.steps += 1
            for e, vel in state.velocities.items():
                if e is not state.scenario.ego:
                    self.v = max(self.v, float(np.linalg.norm(vel[:3])))

        def get_state(self):
            return {"value": self.v, "steps": self.steps}

    seen = []
    g = load_golden("scenarios")
    p = "3fee6507/dt10"
    gym = sga.ScenarioGym(timestep=0.1, state_callbacks=[lambda s: seen.append(s.t)],
                          metrics=[MaxOthersSpeed(), sga.EgoMaxSpeed()])
    gym.set_scenario(_scenario(g, "3fee6507/scenario"))
    gym.rollout()
    m = gym.get_metrics()
    v = g[p + "/vels"][1:, 1:, :3]
    assert m["max_other_speed_steps"] == int(g[p + "/n_steps"])
    assert m["max_other_speed_value"] == np.nanmax(np.linalg.norm(v, axis=-1))
    assert m["ego_max_speed"] == float(g[p + "/metric_ego_max_speed"])
    assert seen[0] == g[p + "/t"][0] and seen[-1] == g[p + "/t"][-1]
    gym.close()

    # StateCallback subclasses (callback.py:9-41): required_callbacks are resolved through state.get_callback at reset
    class Clock(sga.StateCallback):
        def _reset(self, state):
            self.ts = []

        def __call__(self, state):
            self.ts.append(state.t)

    class NeedsClock(sga.StateCallback):
        required_callbacks = [Clock]

        def __call__(self, state):
            self.last = self.callbacks[0].ts[-1]

    clock, dep = Clock(), NeedsClock()
    gym = sga.ScenarioGym(timestep=0.1, state_callbacks=[clock, dep])
    gym.set_scenario(_scenario(g, "3fee6507/scenario"))
    gym.rollout()
    assert bits_equal(np.array(clock.ts), g[p + "/t"]) and dep.last == g[p + "/t"][-1] and dep.callbacks == [clock]
    assert gym.state.get_callback(Clock) is clock and gym.state.get_callback(sga.RSSDistances) is None
    gym.close()
    gym = sga.ScenarioGym(timestep=0.1, state_callbacks=[NeedsClock()])
    with pytest.raises(ValueError, match="Callback Clock is required"):
        gym.set_scenario(_scenario(g, "3fee6507/scenario"))
    gym.close()


def test_batched_gym_with_external_actions_and_mixed_agents():
    """BatchedScenarioGym: four scenes in one batch, ExternalVehicleAgent egos fed [n, R, 2] actions
    (the loop of integrations/openaigym.py:171-226), per-scenario metric dicts."""
    import scenario_gym_amd as sga

    g = load_golden("synth")
    n = int(g["n"])
    scs = [_scenario(g, f"{i}/scenario") for i in range(n)]

    def create_agent(s, e):
        if e.ref == "ego":
            return sga.ExternalVehicleAgent(e)

    gym = sga.BatchedScenarioGym(timestep=0.1, metrics=lambda: [sga.EgoDistanceTravelled(), sga.CollisionMetric()],
                                 event_capacity=64)
    gym.set_scenarios(scs, create_agent=create_agent)
    steps = int(g["0/ext_dt10/n_steps"])
    acts = np.stack([g[f"{i}/ext_dt10/actions"][:steps] for i in range(n)], axis=1)
    gym.step(acts, n=steps)
    ms = gym.get_metrics()
    for i in range(n):
        p = f"{i}/ext_dt10"
        assert abs(ms[i]["ego_distance_travelled"] - float(g[p + "/metric_ego_distance_travelled"])) < 1e-9
        refs = list(g[f"{i}/scenario/refs"])
        assert [(t, refs.index(r)) for t, r, _ in ms[i]["collisions"]] == list(zip(g[p + "/ev_t"], g[p + "/ev_other"]))
        E = len(scs[i].entities)
        st = gym.states[i]
        got = np.array([st.poses.get(e, np.full(6, np.nan)) for e in scs[i].entities])
        assert np.nanmax(np.abs(got - g[p + "/poses"][-1, :E])) < 1e-9
        assert st.is_done
    gym.close()


def test_torch_zero_copy_view():
    """The device state is exposed as raw pointers; torch only wraps them."""
    import scenario_gym_amd as sga
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    packed = synthetic.make_batch(64, 16, n_steps=50)
    eng = sga.RolloutEngine(64, 16)
    eng.upload(packed)
    eng.rollout(50)
    view = eng.torch_state()
    st = eng.state()
    x = view[:, L.F_POSE, :].reshape(-1)[: 64 * 16].reshape(64, 16).cpu().numpy()
    assert np.array_equal(np.where(st["present"], x, np.nan), st["poses"][..., 0], equal_nan=True)
    assert view.is_cuda and view.dtype.is_floating_point and view.shape[1] == L.F_COLL + 1
    eng.close()


def test_pedestrian_agents_through_the_gym_api():
    """PedestrianAgent / SocialForce descriptors through ScenarioGym (tests/pedestrian/test_social_force.py style),
    checked against the reference closed loop."""
    import scenario_gym_amd as sga

    g = load_golden("pedestrian")
    sc = _scenario(g, "loop0/scenario")
    routes, vdes, thr = g["loop0/routes"], g["loop0/vdes"], float(g["loop0/distance_threshold"])
    idx = {e.ref: i for i, e in enumerate(sc.entities)}

    def create_agent(s, e):
        if e.ref == "ego":
            return sga.ReplayTrajectoryAgent(e)
        i = idx[e.ref]
        return sga.PedestrianAgent(e, routes[i], vdes[i], sga.SocialForce(sga.SocialForceParameters(std_lon=0.0, std_lat=0.0)),
                                   distance_threshold=thr)

    gym = sga.ScenarioGym(timestep=0.1, metrics=[sga.CollisionMetric()])
    gym.set_scenario(sc, create_agent=create_agent)
    gym.step()
    first = gym.state.poses
    assert all(np.abs(first[e] - g["loop0/dt10/poses"][1, i]).max() < 1e-9 for e, i in ((e, idx[e.ref]) for e in sc.entities))
    gym.rollout()
    p = "loop0/dt10"
    for e in sc.entities:
        assert np.abs(gym.state.poses[e] - g[p + "/poses"][-1, idx[e.ref]]).max() < 1e-8
    refs = list(g["loop0/scenario/refs"])
    got = gym.get_metrics()["collisions"]
    assert [(t, refs.index(r), ty) for t, r, ty in got] == [(t, int(o), "non_vehicle") for t, o in zip(g[p + "/ev_t"], g[p + "/ev_other"])]
    gym.close()


def test_social_force_with_the_reference_default_noise():
    """SocialForce(SocialForceParameters()) -- the reference's DEFAULT std_lon / std_lat (random_walk.py:13-19) -- runs:
    noise="numpy" reproduces the reference's own rollout after np.random.seed(k) (golden ped_noise/loop2: 70 pedestrians,
    the crowd kernel), the default noise="device" draws the same distribution on the GPU and stays within the noise's
    reach of it."""
    import scenario_gym_amd as sga

    g = load_golden("ped_noise")
    sc = _scenario(g, "loop2/scenario")
    routes, vdes, thr = g["loop2/routes"], g["loop2/vdes"], float(g["loop2/distance_threshold"])
    std_lon, std_lat, seed = g["loop2/noise"]
    assert (std_lon, std_lat) == (sga.SocialForceParameters.std_lon, sga.SocialForceParameters.std_lat)
    idx = {e.ref: i for i, e in enumerate(sc.entities)}
    ref = g["loop2/dt30/poses"]
    for mode in ("numpy", "device"):
        def create_agent(s, e, mode=mode):
            i = idx[e.ref]
            return sga.PedestrianAgent(e, routes[i], vdes[i], sga.SocialForce(sga.SocialForceParameters(noise=mode, noise_seed=int(seed))),
                                       distance_threshold=thr)

        gym = sga.ScenarioGym(timestep=1 / 30, metrics=[sga.CollisionMetric()])
        gym.set_scenario(sc, create_agent=create_agent)
        for _ in range(30):
            gym.step()
        early = np.array([gym.state.poses[e] for e in sc.entities])
        gym.rollout()
        end = np.array([gym.state.poses[e] for e in sc.entities])
        gym.close()
        if mode == "numpy":
            assert np.abs(early - ref[30]).max() < 1e-9 and np.abs(end - ref[-1]).max() < 1e-8
        else:  # another stream of the same tiny noise: different bits, the same walk while the crowd's chaos lets it be
            assert 0 < np.abs(early - ref[30]).max() < 1e-3 and np.isfinite(end).all()


def test_random_walk_agents_through_the_gym():
    """tests/pedestrian/test_random_walk.py: PedestrianAgent(entity, route, speed_desired, behaviour=RandomWalk(params)) steps
    and moves (last speed > 0); with noise="numpy" the gym walks the reference's own rollout after np.random.seed(k) (golden
    random_walk/loop0: nine pedestrians and a car), goal indices included; SocialForce and RandomWalk agents in one gym are
    refused."""
    import scenario_gym_amd as sga

    g = load_golden("random_walk")
    sc = _scenario(g, "loop0/scenario")
    routes, vdes = g["loop0/routes"], g["loop0/vdes"]
    std_lon, std_lat, bias_lon, bias_lat, max_speed, seed = g["loop0/params"]
    idx = {e.ref: i for i, e in enumerate(sc.entities)}
    ref = g["loop0/dt30/poses"]

    def create_agent(s, e):
        if e.ref == "ego":
            return sga.agent._create_agent(s, e)
        i = idx[e.ref]
        params = sga.RandomWalkParameters(std_lon=std_lon, std_lat=std_lat, noise="numpy", noise_seed=int(seed))
        return sga.PedestrianAgent(e, routes[i], vdes[i], sga.RandomWalk(params), max_speed=max_speed)

    gym = sga.ScenarioGym(timestep=1 / 30, metrics=[sga.CollisionMetric()])
    gym.set_scenario(sc, create_agent=create_agent)
    gym.step()
    walker = next(a for a in gym.state.agents.values() if isinstance(a, sga.PedestrianAgent))
    assert walker.speed > 0  # (the reference's test: agent.last_action.speed > 0)
    for _ in range(29):
        gym.step()
    early = np.array([gym.state.poses[e] for e in sc.entities])
    gym.rollout()
    end = np.array([gym.state.poses[e] for e in sc.entities])
    goal = [a.goal_idx for e, a in gym.state.agents.items() if isinstance(a, sga.PedestrianAgent)]
    want = [int(k) for k in g["loop0/dt30/extra"][-1][:, 1] if not np.isnan(k)]
    gym.close()
    assert np.nanmax(np.abs(early - ref[30])) < 1e-9 and np.nanmax(np.abs(end - ref[-1])) < 1e-8 and goal == want



def _mixed_behaviours(sga, g, si, noise="numpy"):
    """The behaviour objects of mixed_peds.npz loop si (one per model) + the model of every entity."""
    cols = [str(c) for c in g["model_cols"]]
    seed = int(g[f"loop{si}/np_seed"])
    out = []
    for row in g[f"loop{si}/models"]:
        m = dict(zip(cols, row))
        if m["behaviour"] == 1:
            out.append(sga.RandomWalk(sga.RandomWalkParameters(bias_lon=m["bias_lon"], bias_lat=m["bias_lat"], std_lon=m["std_lon"],
                                                               std_lat=m["std_lat"], max_speed_factor=m["max_speed_factor"],
                                                               noise=noise, noise_seed=seed)))
        else:
            out.append(sga.SocialForce(sga.SocialForceParameters(
                relaxation_time=m["relaxation_time"], ped_repulse_V=m["ped_repulse_V"], ped_repulse_sigma=m["ped_repulse_sigma"],
                ped_attract_C=m["ped_attract_C"], sight_weight=m["sight_weight"], sight_weight_use=bool(m["sight_weight_use"]),
                sight_angle=m["sight_angle"], max_speed_factor=m["max_speed_factor"], bias_lon=m["bias_lon"], bias_lat=m["bias_lat"],
                std_lon=m["std_lon"], std_lat=m["std_lat"], noise=noise, noise_seed=seed)))
    return out, g[f"loop{si}/model_of"]


@pytest.mark.parametrize("si", [0, 1, 2, 3])
def test_every_pedestrian_agent_with_its_own_behaviour(si):
    """pedestrian/agent.py:18-41: a PedestrianAgent holds its OWN behaviour object -- SocialForce pedestrians of two or three
    parameter sets and RandomWalk pedestrians in one scenario.  The gym lowers the distinct (behaviour, parameters, std)
    combinations to device models (sg_set_ped_models) and walks the reference's own closed loop after np.random.seed(k)
    (mixed_peds.npz, generated from the real reference): poses after 30 steps and at the end, goal indices, and the force
    every SocialForce pedestrian felt last -- RandomWalk pedestrians never touch theirs (random_walk.py:37-43)."""
    import scenario_gym_amd as sga

    g = load_golden("mixed_peds")
    sc = _scenario(g, f"loop{si}/scenario")
    routes, vdes = g[f"loop{si}/routes"], g[f"loop{si}/vdes"]
    behaviours, model_of = _mixed_behaviours(sga, g, si)
    idx = {e.ref: i for i, e in enumerate(sc.entities)}
    ref, ex = g[f"loop{si}/dt30/poses"], g[f"loop{si}/dt30/extra"]

    def create_agent(s, e):
        if e.ref == "ego":
            return sga.agent._create_agent(s, e)
        i = idx[e.ref]
        return sga.PedestrianAgent(e, routes[i], vdes[i], behaviours[model_of[i]])

    gym = sga.ScenarioGym(timestep=1 / 30, metrics=[sga.CollisionMetric()])
    gym.set_scenario(sc, create_agent=create_agent)
    for _ in range(30):
        gym.step()
    early = np.array([gym.state.poses[e] for e in sc.entities])
    gym.rollout()
    end = np.array([gym.state.poses[e] for e in sc.entities])
    peds = [(idx[e.ref], a) for e, a in gym.state.agents.items() if isinstance(a, sga.PedestrianAgent)]
    goal = [a.goal_idx for _, a in peds]
    force = np.array([a.force for _, a in peds])
    want_goal = [int(ex[-1][i, 1]) for i, _ in peds]
    want_force = np.array([ex[-1][i, 2:] for i, _ in peds])
    n_models = gym._b.engine_models()
    gym.close()
    assert np.nanmax(np.abs(early - ref[30])) < 1e-9 and np.nanmax(np.abs(end - ref[-1])) < 1e-8
    assert goal == want_goal and np.abs(force - want_force).max() < 1e-8
    assert n_models == len(behaviours)


def test_ped_models_on_the_device_equal_the_oracle(oracle):
    """sg_set_ped_models through the engine on a batch: 40 scenarios x 48 pedestrians (general pedestrian variant, one
    wavefront per tile), 6 x 200 (four wavefronts per scenario) and 2 x 600 (the multi-kernel step) with three models dealt out at random -- two SocialForce
    parameter sets and a RandomWalk -- with the counter-based noise: final state, forces, metrics and events equal the oracle's
    per-agent models bit for bit; and the same batch under ONE model differs (the models do act)."""
    import scenario_gym_amd as sga
    from oracle import check
    from scenario_gym_amd import synthetic

    models = [dict(std_lon=0.05, std_lat=0.02),
              dict(relaxation_time=0.8, ped_repulse_V=2.5, ped_repulse_sigma=0.6, sight_weight=0.3, sight_angle=160,
                   max_speed_factor=1.1, bias_lon=0.05, bias_lat=-0.02, std_lon=0.02, std_lat=0.1),
              dict(behaviour="random_walk", bias_lon=0.1, bias_lat=0.05, std_lon=0.3, std_lat=0.2)]
    rows = []
    for m in models:
        d = {k: v for k, v in m.items() if k not in ("behaviour", "std_lon", "std_lat")}
        rows.append(oracle.ped_model_row(m.get("behaviour", "social_force"), oracle.social_force_params(**d), m["std_lon"], m["std_lat"]))
    rows = np.array(rows)
    for R, E, side, steps in ((40, 48, 14.0, 300), (6, 200, 30.0, 250), (2, 600, 45.0, 60)):  # (600: the multi-kernel step)
        dt = 1 / 30
        packed = synthetic.make_crowd(R, E, n_steps=steps, side=side)
        model_of = np.random.default_rng(R).integers(0, 3, R * E).astype(np.int32)
        finals = []
        for mo in (model_of, None):
            eng = sga.RolloutEngine(R, E, timestep=dt, terminal_conditions=["max_length"], event_capacity=128)
            eng.set_ped_models(models if mo is not None else models[:1], mo, noise="device", noise_seed=7)
            eng.upload(packed)
            eng.rollout(steps)
            st = eng.state()
            rws, evs = eng.metrics()
            finals.append(st["poses"].copy())
            if mo is not None:
                for r in check.spread(R, 5):
                    o = check.oracle_final(packed, r, dt, steps, event_cap=128,
                                           noise=dict(mode="device", std_lon=0.0, std_lat=0.0, seed=7, scenario_index=r),
                                           models=rows, model_of=mo[r * E:(r + 1) * E])
                    bad = check.compare_final(st, rws, evs, r, o, E, event_cap=128, ped=True, kind=packed.kind[r * E:(r + 1) * E])
                    assert not bad, (R, E, r, bad)
            eng.close()
        assert np.nanmax(np.abs(finals[0] - finals[1])) > 1e-2


def test_crowd_kernel_with_several_pedestrian_models(oracle, monkeypatch):
    """All-pedestrian batches whose pedestrians follow several SocialForce parameter sets (PedestrianAgent(entity, route, speed,
    behaviour): pedestrian/agent.py:18-41) stay on the crowd kernel: a pass of the force code per model with that model's
    constants (rollout_kernel_crowd_models<2 | 4>).  Same bits as the general pedestrian variant (SG_CROWD_MODELS=0) and as the
    oracle's per-agent models, with the counter-based noise; a model with an attraction term falls back to the general pair code
    inside the crowd kernel; a RandomWalk among the models sends the batch to the general variant."""
    import scenario_gym_amd as sga
    from oracle import check
    from scenario_gym_amd import synthetic

    models = [dict(std_lon=0.05, std_lat=0.02),
              dict(relaxation_time=0.8, ped_repulse_V=2.5, ped_repulse_sigma=0.6, sight_weight=0.3, sight_angle=160,
                   max_speed_factor=1.1, bias_lon=0.05, bias_lat=-0.02, std_lon=0.02, std_lat=0.1),
              dict(ped_repulse_V=1.4, ped_attract_C=0.002, std_lon=0.0, std_lat=0.0),  # (attraction: outside crowd_pair's shortcuts)
              dict(relaxation_time=0.3, ped_repulse_sigma=0.35, sight_weight_use=False, std_lon=0.01, std_lat=0.01)]
    rows = []
    for m in models:
        d = {k: v for k, v in m.items() if k not in ("behaviour", "std_lon", "std_lat")}
        rows.append(oracle.ped_model_row("social_force", oracle.social_force_params(**d), m["std_lon"], m["std_lat"]))
    rows = np.array(rows)
    dt = 1 / 30
    for R, E, side, steps in ((8, 120, 22.0, 260), (6, 230, 32.0, 240)):
        packed = synthetic.make_crowd(R, E, n_steps=steps, side=side)
        model_of = np.random.default_rng(E).integers(0, len(models), R * E).astype(np.int32)
        out = []
        for crowd in ("1", "0"):
            monkeypatch.setenv("SG_CROWD_MODELS", crowd)
            eng = sga.RolloutEngine(R, E, timestep=dt, terminal_conditions=["max_length"], event_capacity=128)
            eng.set_ped_models(models, model_of, noise="device", noise_seed=3)
            eng.upload(packed)
            eng.rollout(steps)
            st = eng.state()
            rws, evs = eng.metrics()
            out.append((st, rws, evs, eng.last_kernel()))
            if crowd == "1":
                for r in range(R):
                    o = check.oracle_final(packed, r, dt, steps, event_cap=128, noise=dict(mode="device", std_lon=0.0, std_lat=0.0, seed=3, scenario_index=r),
                                           models=rows, model_of=model_of[r * E:(r + 1) * E])
                    bad = check.compare_final(st, rws, evs, r, o, E, event_cap=128, ped=True, kind=packed.kind[r * E:(r + 1) * E])
                    assert not bad, (E, r, bad)
            eng.close()
        (sa, ra, ea, ka), (sb, rb, eb, kb) = out
        assert ka == f"sg::rollout_kernel_crowd_models<{2 if E <= 128 else 4}>" and kb == f"sg::rollout_kernel<64, {2 if E <= 128 else 4}, true, false>", (ka, kb)
        for k in ("poses", "vels", "dists", "force", "ctrl_state"):
            assert bits_equal(sa[k], sb[k]), k
        assert np.array_equal(sa["coll"], sb["coll"]) and ra.tobytes() == rb.tobytes() and ea.tobytes() == eb.tobytes()
    eng = sga.RolloutEngine(4, 100, timestep=dt)
    eng.set_ped_models(models[:1] + [dict(behaviour="random_walk", std_lon=0.1, std_lat=0.1)], np.zeros(400, np.int32), noise="device")
    eng.upload(synthetic.make_crowd(4, 100, n_steps=30, side=20.0))
    eng.rollout(30)
    assert eng.last_kernel() == "sg::rollout_kernel<64, 2, true, false>"
    eng.close()


def test_scenario_of_700_entities_through_the_gym(oracle):
    """A Scenario object of 700 entities through the reference's one-scenario API (no entity ceiling: state/utils.py:10-49
    has none): set_scenario -> rollout; State.poses / collisions() (rows of eleven words) / the metrics equal the oracle's."""
    import scenario_gym_amd as sga
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic
    from scenario_gym_amd.packing import unpack_scenario

    E, steps = 700, 60
    packed = synthetic.make_batch(1, E, n_steps=steps, ego_kind=L.KIND_AGENT_REPLAY, static_frac=0.2, vanish_frac=0.1, extent=110.0, n_knots=10)
    s = unpack_scenario(packed, 0)
    sc = scenario_from_arrays(s, [f"entity_{k}" if k else "ego" for k in range(E)])
    gym = sga.ScenarioGym(timestep=1 / 30, metrics=[sga.CollisionMetric(), sga.EgoDistanceTravelled()])
    gym.set_scenario(sc)
    gym.rollout()
    s = unpack_scenario(gym._b._packed, 0)  # (what the gym packed: Trajectory() has normalised the knots -- headings unwrapped)
    o = oracle.rollout(s["knot_off"], s["knots"], s["bbox"], s["etype"], s["kind"], s["ego"], s["t0"], s["length"], 1 / 30,
                       ctrl=s["ctrl"], event_cap=256)
    st = gym.state
    assert st.t == o["final_t"]
    ents = sc.entities
    dense = oracle.coll_to_dense(o["coll"], E)[-1]
    got = st.collisions()
    n_pairs = 0
    for i, e in enumerate(ents):
        want = [ents[j] for j in np.flatnonzero(dense[i])]
        assert got.get(e, []) == want, i
        n_pairs += len(want)
        if e in st.poses:
            assert bits_equal(st.poses[e], o["poses"][-1, i]), i
    assert n_pairs > 0
    m = gym.get_metrics()
    assert m["ego_distance_travelled"] == o["metric_ego_distance_travelled"]
    assert [(t, r) for t, r, _ in m["collisions"]] == [(t, ents[int(j)].ref) for t, j in zip(o["ev_t"], o["ev_other"])]
    gym.close()


def test_terminal_conditions_mapping():
    """state.py:397-408: TERMINAL_CONDITIONS[name](state) -- the four predicates a caller (e.g. the RL reward of
    integrations/openaigym.py:300-310) evaluates on a state, whatever conditions the gym itself stops on: on one of the
    reference's scenarios, step by step, they equal the reference's own expressions over the State (max_length at the end;
    the Scenario object carries no road network here: the ego counts as off the road)."""
    import scenario_gym_amd as sga

    g = load_golden("scenarios")
    sc = _scenario(g, "a5e43fe4/scenario")
    gym = sga.ScenarioGym(timestep=0.1, terminal_conditions=["max_length"])
    gym.set_scenario(sc)
    assert set(sga.TERMINAL_CONDITIONS) == {"max_length", "collision", "ego_collision", "ego_off_road"}
    seen_done = False
    while not gym.state.is_done:
        s = gym.state
        assert sga.TERMINAL_CONDITIONS["max_length"](s) == (s.t + s.dt > s.scenario.length)
        assert sga.TERMINAL_CONDITIONS["collision"](s) == any(len(v) > 0 for v in s.collisions().values())
        assert sga.TERMINAL_CONDITIONS["ego_collision"](s) == (len(s.collisions()[s.scenario.entities[0]]) > 0)
        gym.step()
        seen_done = True
    assert seen_done and sga.TERMINAL_CONDITIONS["max_length"](gym.state)
    assert sga.TERMINAL_CONDITIONS["ego_off_road"](gym.state)  # (no road network attached to this Scenario object: off the road)
    gym.close()


def test_combined_sensor():
    """tests/test_sensor.py:13-36: CombinedSensor over the localisation, future-collision and collision sensors of the ego
    (the keyboard sensor needs the viewer: out of scope) -- no observation class before the reset, one afterwards, and an
    observation that carries every sensor's fields under the reference's names; the values are the single sensors'."""
    import scenario_gym_amd as sga

    g = load_golden("scenarios")
    gym = sga.ScenarioGym()
    gym.set_scenario(_scenario(g, "a5e43fe4/scenario"))
    ego = gym.state.scenario.entities[0]
    state = gym.state
    sensor = sga.CombinedSensor(ego, sga.EgoLocalizationSensor(ego), sga.FutureCollisionDetector(ego), sga.GlobalCollisionDetector(ego),
                                sga.RasterizedMapSensor(ego, layers=["entity"], n=16))
    assert sensor.obs_class is None
    sensor.reset(state)
    assert sensor.obs_class is not None
    gym.step()
    obs = sensor.step(state)
    assert obs.entity is ego and obs.t == state.t and obs.next_t == state.next_t
    assert np.array_equal(obs.pose, state.poses[ego]) and np.array_equal(obs.velocity, state.velocities[ego])
    assert obs.future_collision == sga.FutureCollisionDetector(ego).step(state).future_collision
    assert obs.collisions == state.collisions() and obs.map.shape == (16, 16, 1)
    names = [f.name for f in __import__("dataclasses").fields(obs)]
    assert names == ["entity", "t", "next_t", "pose", "velocity", "distance_travelled", "recorded_poses", "entity_state",
                     "future_collision", "collisions", "map"]
    # duplicate field names: skipped without prefixes, kept under a prefix with them
    C = sga.combine_observations(sga.SingleEntityObservation, sga.FutureCollisionObservation, prefixes=("a", "b"))
    assert [f.name for f in __import__("dataclasses").fields(C)][8:] == ["b_entity", "b_t", "b_next_t", "b_pose", "b_velocity",
                                                                         "b_distance_travelled", "b_recorded_poses", "b_entity_state",
                                                                         "future_collision"]
    gym.close()


def test_to_scenario_round_trip():
    """tests/test_state.py:210-260: roll out, write the recording back as a scenario (State.to_scenario), roll the
    recording out again: same entities, the ego follows the recorded poses."""
    import scenario_gym_amd as sga
    from scenario_gym_amd.trajectory import is_stationary

    g = load_golden("scenarios")
    scenario = _scenario(g, "a5e43fe4/scenario").reset_start()
    gym = sga.ScenarioGym()
    gym.set_scenario(scenario)
    gym.rollout()
    poses = gym.state.recorded_poses()[scenario.entities[0]]
    assert np.unique(poses, axis=0).shape[0] == poses.shape[0]
    new_scenario = gym.state.to_scenario()
    ego = new_scenario.entities[0]
    assert len(ego.trajectory.t) == ego.trajectory.data.shape[0] == poses.shape[0]
    assert len(new_scenario.entities) == len(scenario.entities)
    assert all(type(a) is type(b) and a.ref == b.ref for a, b in zip(scenario.entities, new_scenario.entities))
    for old, new in zip(scenario.entities, new_scenario.entities):
        assert (len(new.trajectory) == 1) == is_stationary(gym.state.recorded_poses()[old])
    d = ego.trajectory.data  # Trajectory.__init__ normalises (heading unwrap): positions and times stay bit-identical
    assert bits_equal(d[:, :4], poses[:, :4]) and np.abs(np.angle(np.exp(1j * (d[:, 4] - poses[:, 4])))).max() < 1e-12
    gym2 = sga.ScenarioGym()
    gym2.set_scenario(new_scenario)
    gym2.rollout()
    again = gym2.state.recorded_poses()[new_scenario.entities[0]]
    n = min(len(again), len(poses))
    assert n >= len(poses) - 1 and bits_equal(again[:n, 0], poses[:n, 0])      # same clock
    assert np.abs(again[:n, 1:4] - poses[:n, 1:4]).max() < 1e-9                # knots are the recorded poses
    gym.close()
    gym2.close()


def test_entities_in_area_and_radius():
    """tests/test_state.py:82-97 (radius counts) and state.py:340-354 with a polygon given by its vertices."""
    import scenario_gym_amd as sga

    g = load_golden("scenarios")
    gym = sga.ScenarioGym()
    gym.set_scenario(_scenario(g, "a5e43fe4/scenario"))
    gym.reset_scenario()
    gym.step()
    st = gym.state
    ego = st.scenario.entities[0]
    pose = st.poses[ego]
    others = {e: p for e, p in st.poses.items() if e is not ego}
    d = np.array([np.hypot(*(p[:2] - pose[:2])) for p in others.values()])
    assert len(st.get_entities_in_radius(*pose[:2], d.min() - 0.1)) == 1
    # Point.buffer(r) is the 64-gon INSCRIBED in the circle: its apothem is r cos(pi/64)
    assert len(st.get_entities_in_radius(*pose[:2], (d.max() + 1) / np.cos(np.pi / 64))) == len(st.poses)
    x, y = pose[:2]
    square = np.array([[x - 1, y - 1], [x + 1, y - 1], [x + 1, y + 1], [x - 1, y + 1]])
    inside = st.get_entities_in_area(square)
    assert ego in inside and all(abs(st.poses[e][0] - x) < 1 and abs(st.poses[e][1] - y) < 1 for e in inside)
    assert len(st.get_entities_in_area(square + 1e6)) == 0
    gym.close()


def test_state_info_radius_counts_as_the_reference_asserts():
    """tests/test_state.py:75-97, statement by statement, on the scenario the reference's fixture names (3e39a079...): timestep
    0.1, 50 steps, then get_entities_in_radius around entities[0] with min(distance) - 0.1 finds exactly 1 entity and with
    max(distance) + 1 all of them (distances are the 3-d norms of the reference's assertion; the test evaluates the 64-gon
    Point.buffer(r) of state.py:356-372)."""
    import json

    import scenario_gym_amd as sga
    from scenario_gym_amd.scenario import Scenario

    gj = load_golden("json")
    n = "3e39a079-5653-440c-bcbe-24dc9f6bf0e6"
    d = json.loads(str(gj[f"{n}/to_dict"]))
    for e in d["entities"]:
        e["trajectory"] = gj[f"{n}/traj_{e['trajectory']}"].tolist()
    d["road_network"] = None
    gym = sga.ScenarioGym(timestep=0.1)
    gym.set_scenario(Scenario.from_dict(d))
    for _ in range(50):
        gym.step()
    st = gym.state
    assert len(st.poses) >= 2
    e = st.scenario.entities[0]
    pose = st.poses[e]
    distances = [np.linalg.norm(p[:3] - pose[:3]) for e_, p in st.poses.items() if e_ is not e]
    assert len(st.get_entities_in_radius(*pose[:2], np.min(distances) - 0.1)) == 1
    assert len(st.get_entities_in_radius(*pose[:2], np.max(distances) + 1)) == 1 + len(distances)
    gym.close()


def _actions_scenario(g, tag):
    """The reference's 1518e754... scenario (json.npz) with the actions the reference stepped it with (actions.npz)."""
    import json

    from scenario_gym_amd import actions as A
    from scenario_gym_amd.scenario import Scenario

    gj = load_golden("json")
    n = str(g["name"])
    d = json.loads(str(gj[f"{n}/to_dict"]))
    for e in d["entities"]:
        e["trajectory"] = gj[f"{n}/traj_{e['trajectory']}"].tolist()
    d["road_network"] = None
    s = Scenario.from_dict(d)
    by_name = {"UserDefinedAction": A.UserDefinedAction, "UpdateStateVariableAction": A.UpdateStateVariableAction}
    s.actions = [by_name[str(c)].from_dict(a) for c, a in zip(g[f"{tag}/classes"], json.loads(str(g[f"{tag}/to_dict_actions"])))]
    return s


@pytest.mark.parametrize("tag,dt", [("dt30", 1.0 / 30.0), ("dt10", 0.1)])
def test_scenario_actions_match_reference(tag, dt):
    """State.update_actions / apply_action / entity_state / action_apply_times (state/state.py:150-160, 241-266;
    scenario/actions.py:12-168) against the reference stepped through the same scenario (tests/golden/make_golden_actions.py):
    a UserDefinedAction from the OpenSCENARIO file (trigger t >= action time) and UpdateStateVariableActions (trigger t >
    action time) before the start, exactly on a step time, between steps, for an unknown entity, after the end.  Tick by
    tick: the application time of every action bit for bit, the ego's entity state after every step, the final entity
    states, what stays unapplied.  Then the whole rollout as ONE device call: the same times and states from the clock."""
    import json
    import warnings

    import scenario_gym_amd as sga

    g = load_golden("actions")
    s = _actions_scenario(g, tag)
    assert [a.t for a in s.actions] == list(g[f"{tag}/t"]) and [a.entity_ref for a in s.actions] == [str(x) for x in g[f"{tag}/entity_ref"]]
    assert json.dumps([a.to_dict() for a in s.actions]) == str(g[f"{tag}/to_dict_actions"])       # key order too
    assert bits_equal([a.t for a in s.reset_start().actions], g[f"{tag}/t_after_reset_start"])    # FixedTAction.translate
    acts = list(s.actions)
    gym = sga.ScenarioGym(timestep=dt)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # (apply_action warns about the entity nobody knows, as the reference does)
        gym.set_scenario(s)
        assert bits_equal([gym.state.action_apply_times[a] for a in acts], g[f"{tag}/apply_times_after_reset"])
        per_step, clock = [json.dumps(gym.state.get_entity_data(s.ego)[-1], sort_keys=True)], [gym.state.t]
        while not gym.state.is_done:
            gym.step()
            per_step.append(json.dumps(gym.state.get_entity_data(s.ego)[-1], sort_keys=True))
            clock.append(gym.state.t)
    assert bits_equal(clock, g[f"{tag}/clock"])
    assert per_step == [str(x) for x in g[f"{tag}/ego_state_per_step"]]
    assert bits_equal([gym.state.action_apply_times[a] for a in acts], g[f"{tag}/apply_times"])
    assert json.dumps({e.ref: gym.state.entity_state[e] for e in s.entities}, sort_keys=True) == str(g[f"{tag}/entity_state"])
    assert len(gym.state.unapplied_actions) == int(g[f"{tag}/n_unapplied"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gym.rollout()  # reset + every step in one device call; the actions are replayed from the clock
    assert bits_equal([gym.state.action_apply_times[a] for a in acts], g[f"{tag}/apply_times"])
    assert json.dumps({e.ref: gym.state.entity_state[e] for e in s.entities}, sort_keys=True) == str(g[f"{tag}/entity_state"])
    assert len(gym.state.unapplied_actions) == int(g[f"{tag}/n_unapplied"])

    class Late(sga.ScenarioAction):  # the caller's own action class: evaluated tick by tick (the per-step host path)
        def trigger_condition(self, state):
            return len(state.poses) >= 2 and state.t > s.ego.trajectory.min_t + 1.0

        def _apply(self, state, entity):
            state.entity_state[entity] = {"seen": len(state.poses)}

    s2 = s.add_action(Late("Late", s.ego.ref, {}))
    gym.set_scenario(s2)
    gym.rollout()
    late = s2.actions[-1]
    t_applied = gym.state.action_apply_times[late]
    assert t_applied > s.ego.trajectory.min_t + 1.0 and t_applied - dt <= s.ego.trajectory.min_t + 1.0 + 1e-12
    assert gym.state.entity_state[gym.state.scenario.ego]["seen"] >= 2
    gym.close()


# --------------------------------------------------------------------------- caller-run (Python) agents
def _python_replay_agent(sga):
    class PyReplayAgent(sga.Agent):
        """ReplayTrajectoryAgent (agent.py:118-128) written as a user would: a Python _step."""

        def __init__(self, entity):
            super().__init__(entity, sga.ReplayTrajectoryController(entity), sga.EgoLocalizationSensor(entity))
            self.calls = 0

        def _step(self, observation):
            t, next_t = observation.t, observation.next_t
            self.calls += 1
            return sga.TeleportAction(pose=self.entity.trajectory.position_at_t(next_t))

    return PyReplayAgent


def test_python_agent_equals_device_replay_agent():
    """An Agent subclass with a Python _step (sensor -> _step -> controller on the host, pose injected into the device
    step) reproduces the built-in ReplayTrajectoryAgent bit for bit: same poses, velocities, metrics, collisions."""
    import scenario_gym_amd as sga

    g = load_golden("scenarios")
    Py = _python_replay_agent(sga)
    made = []

    def create_py(scenario, entity):
        if entity.ref == "ego":
            made.append(Py(entity))
            return made[-1]

    out = []
    for create in (None, create_py):
        gym = sga.ScenarioGym(metrics=[sga.EgoAvgSpeed(), sga.EgoMaxSpeed(), sga.EgoDistanceTravelled(), sga.CollisionMetric()])
        kw = {} if create is None else dict(create_agent=create)
        gym.set_scenario(_scenario(g, "a5e43fe4/scenario"), **kw)
        gym.rollout()
        st = gym.state
        out.append((st.recorded_poses(), st.velocities, st.distances, gym.get_metrics(), st.t))
        gym.close()
    (pa, va, da, ma, ta), (pb, vb, db, mb, tb) = out
    assert made and made[0].calls > 100 and ta == tb
    for (ea, ra), (eb, rb) in zip(pa.items(), pb.items()):
        assert ea.ref == eb.ref and bits_equal(ra, rb), ea.ref
    for (ea, xa), (eb, xb) in zip(va.items(), vb.items()):
        assert bits_equal(xa, xb) and da[ea] == db[eb]
    assert ma == mb
    assert bits_equal(pa[next(iter(pa))][:, 1:], g["a5e43fe4/dt30/poses"][:, 0][~np.isnan(g["a5e43fe4/dt30/poses"][:, 0, 0])])


@pytest.mark.parametrize("persist", [False, True])
def test_python_agent_returning_none(persist):
    """scenario_gym.py:233-239: an agent that returns None vanishes, or keeps its pose under persist."""
    import scenario_gym_amd as sga

    g = load_golden("scenarios")

    class Quitter(sga.Agent):
        def __init__(self, entity):
            super().__init__(entity, sga.ReplayTrajectoryController(entity), sga.EgoLocalizationSensor(entity))

        def step(self, state):  # overriding step itself is allowed too
            if state.t > self.entity.trajectory.min_t + 1.0:
                return None
            return self.entity.trajectory.position_at_t(state.next_t)

    gym = sga.ScenarioGym(persist=persist)
    gym.set_scenario(_scenario(g, "a5e43fe4/scenario"), create_agent=lambda sc, e: Quitter(e) if e.ref == "ego" else None)
    gym.reset_scenario()
    ego = gym.state.scenario.entities[0]
    last = None
    for _ in range(60):
        if ego in gym.state.poses:
            last = gym.state.poses[ego].copy()
        gym.step()
    if persist:
        assert ego in gym.state.poses and bits_equal(gym.state.poses[ego], last)
        assert np.all(gym.state.velocities[ego] == 0)
    else:
        assert ego not in gym.state.poses
    gym.close()


def test_python_policy_over_device_vehicle_controller():
    """A Python policy (`_step` -> VehicleAction) paired with the built-in VehicleController: the policy runs in the
    caller every tick, the controller stays on the device -- the same bits as ExternalVehicleAgent fed the same actions."""
    import scenario_gym_amd as sga
    import scenario_gym_amd._lib as L

    g = load_golden("scenarios")
    acts = np.random.default_rng(3).uniform([-2, -0.3], [2, 0.3], (50, 2))

    class Policy(sga.Agent):
        def __init__(self, entity):
            super().__init__(entity, sga.VehicleController(entity), sga.EgoLocalizationSensor(entity))
            self.k = 0

        def _step(self, observation):
            assert observation.pose is not None and observation.entity is self.entity  # SingleEntityObservation
            a = sga.VehicleAction(*acts[self.k])
            self.k += 1
            return a

    made = []
    gym = sga.ScenarioGym()
    gym.set_scenario(_scenario(g, "a5e43fe4/scenario"),
                     create_agent=lambda sc, e: (made.append(Policy(e)) or made[-1]) if e.ref == "ego" else None)
    assert made[0].device_kind() == L.KIND_AGENT_VEHICLE
    gym.reset_scenario()
    ref = sga.ScenarioGym()
    ref.set_scenario(_scenario(g, "a5e43fe4/scenario"),
                     create_agent=lambda sc, e: sga.ExternalVehicleAgent(e) if e.ref == "ego" else None)
    ref.reset_scenario()
    for k in range(50):
        gym.step()
        ref.step(acts[k])
    a, b = gym.state.poses[gym.state.scenario.entities[0]], ref.state.poses[ref.state.scenario.entities[0]]
    assert made[0].k == 50 and bits_equal(a, b) and np.abs(a[:2] - g["a5e43fe4/scenario/knots"][0, 1:3]).max() > 0.1
    gym.close()
    ref.close()


def test_batched_run_scenarios_from_files(tmp_path):
    """scenario_gym.py:16-27 (`ScenarioGym.run_scenarios(paths)`) as one device batch: every file's metrics equal
    the one-scenario-at-a-time ScenarioGym run."""
    import scenario_gym_amd as sga
    from test_host_api import CATALOG, XOSC

    (tmp_path / "cats").mkdir()
    (tmp_path / "cats" / "c.xosc").write_text(CATALOG)
    paths = []
    for k in range(5):  # the same small scenario with the ego's middle vertex moved
        f = tmp_path / f"s{k}.xosc"
        f.write_text(XOSC.replace('x="5"', f'x="{5 + k}"'))
        paths.append(str(f))
    metrics = lambda: [sga.EgoAvgSpeed(), sga.EgoMaxSpeed(), sga.EgoDistanceTravelled(), sga.CollisionMetric()]
    batch = sga.BatchedScenarioGym.run_scenarios(paths, metrics=metrics, timestep=0.05)
    assert len(batch) == 5 and len({m["ego_distance_travelled"] for m in batch}) == 5
    for f, mb in zip(paths, batch):
        gym = sga.ScenarioGym(timestep=0.05, metrics=metrics())
        gym.load_scenario(f, relabel=True)
        gym.rollout()
        assert gym.get_metrics() == mb, f
        gym.close()


def test_future_collision_detector_matches_reference():
    """FutureCollisionDetector (sensor/common.py:59-106) evaluated on the device after reset and after every step of the
    reference's own rollouts (five XOSC scenarios, two time steps, horizons 5.0 and 1.0): every flag equals the
    reference's; the sensor object returns it inside the observation."""
    import scenario_gym_amd as sga

    gs, g = load_golden("scenarios"), load_golden("sensors")
    n_pos = 0
    for name in g["names"]:
        for dtn, dt in (("dt30", 1 / 30), ("dt10", 0.1)):
            ts, want = g[f"{name}/{dtn}/t"], g[f"{name}/{dtn}/future"].astype(bool)
            gym = sga.ScenarioGym(timestep=dt)
            gym.set_scenario(_scenario(gs, f"{name}/scenario"))
            gym.reset_scenario()
            sensor = sga.FutureCollisionDetector(gym.state.scenario.entities[0], horizon=5.0)
            for k, t in enumerate(ts):
                assert gym.state.t == t
                got = [gym.state.future_collision(h) for h in g["horizons"]]
                assert got == list(want[k]), (name, dtn, k)
                if k % 50 == 0:
                    assert sensor.step(gym.state).future_collision == want[k, 0]
                n_pos += sum(got)
                if k + 1 < len(ts):
                    gym.step()
            gym.close()
    assert n_pos > 300


def test_raster_entity_layer_matches_reference():
    """RasterizedMapSensor(ego, layers=["entity"]) (sensor/map.py:120-192) on the device, on every 4th state of the
    reference's own dt = 0.1 rollouts of five XOSC scenarios, in both recorded grid configurations: every cell equals the
    reference's; the sensor object returns the map inside the observation."""
    import scenario_gym_amd as sga

    gs, g = load_golden("scenarios"), load_golden("sensors")
    ones = 0
    for name in g["names"]:
        gym = sga.ScenarioGym(timestep=0.1)
        gym.set_scenario(_scenario(gs, f"{name}/scenario"))
        gym.reset_scenario()
        ego = gym.state.scenario.entities[0]
        w0, h0, n0 = g["raster_cfg"][0]
        sensor = sga.RasterizedMapSensor(ego, layers=["entity"], width=w0, height=h0, freq=None, n=int(n0))
        steps = list(g[f"{name}/dt10/map_steps"])
        for k in range(int(steps[-1]) + 1):
            if k in steps:
                f = steps.index(k)
                for c, (w, h, n) in enumerate(g["raster_cfg"]):
                    got = gym.state.entity_raster(w, h, int(n), int(n))
                    want = g[f"{name}/dt10/map{c}"][f].astype(bool)
                    assert np.array_equal(got, want), (name, c, k, int((got != want).sum()))
                    ones += int(got.sum())
                if f % 10 == 0:
                    m = sensor.step(gym.state).map
                    assert m.shape == (int(n0), int(n0), 1) and np.array_equal(m[:, :, 0], g[f"{name}/dt10/map0"][f].astype(bool))
            gym.step()
        gym.close()
    assert ones > 10000


def test_all_reference_scenarios_as_one_batch():
    """The reference's tests/test_scenarios.py rolls out each of its 23 OpenSCENARIO files in turn; here they are ONE
    ragged device batch (1 ... 9 entities, different lengths and start times): clock, poses of every 25th step and of the
    last, final velocities / distances / collisions and the ego metrics equal the real reference's, bit for bit."""
    import scenario_gym_amd as sga

    g = load_golden("all_scenarios")
    names = list(g["names"])
    scs = [scenario_from_arrays(scenario_arrays(g, f"{n}/scenario"), g[f"{n}/scenario/refs"]) for n in names]
    gym = sga.BatchedScenarioGym(record=True, metrics=lambda: [sga.EgoAvgSpeed(), sga.EgoMaxSpeed(), sga.EgoDistanceTravelled()])
    gym.set_scenarios(scs)
    gym.rollout()
    metrics = gym.get_metrics()
    for i, n in enumerate(names):
        st = gym.states[i]
        ents = st.scenario.entities
        ts = g[f"{n}/t"]
        rec = st.recorded_poses()
        assert st.t == ts[-1] and st.is_done, n
        final = g[f"{n}/final_poses"]
        for k, e in enumerate(ents):
            present = not np.isnan(final[k, 0])
            assert (e in st.poses) == present, (n, e.ref)
            if present:
                assert bits_equal(st.poses[e], final[k]) and bits_equal(st.velocities[e], g[f"{n}/final_vels"][k]), (n, e.ref)
            assert st.distances[e] == g[f"{n}/final_dists"][k], (n, e.ref)
            # every 25th recorded step of the entity: rows [t, pose] of the steps it was present
            kf = g[f"{n}/keyframes"][:, k]
            want = np.array([np.concatenate([[ts[25 * j]], kf[j]]) for j in range(len(kf)) if not np.isnan(kf[j, 0])]).reshape(-1, 7)
            got = rec[e][np.isin(rec[e][:, 0], ts[::25])]
            assert bits_equal(got, want), (n, e.ref)
        coll = st.collisions()
        A = g[f"{n}/final_coll"]
        for k, e in enumerate(ents):
            if e in coll:
                assert sorted(ents.index(o) for o in coll[e]) == list(np.nonzero(A[k])[0]), (n, e.ref)
        for key in ("ego_avg_speed", "ego_max_speed", "ego_distance_travelled"):
            assert metrics[i][key] == float(g[f"{n}/metric_{key}"]), (n, key)
    gym.close()


def _road_network_from_golden(g, name):
    """A RoadNetwork carrying the exported polygon arrays (the JSON files do not travel to the GPU box)."""
    from scenario_gym_amd.road_network import RoadNetwork

    rn = RoadNetwork(name=name)
    rn._arrays = {k: g[f"net/{name}/{k}"] for k in ("ring_off", "vert_off", "verts", "layers")}
    return rn


def test_map_sensor_and_ego_off_road_through_the_gym():
    """tests/test_sensor.py:38-77 (RasterizedMapSensor, default layers and all layers, 61 x 61 over 30 m) and a gym with
    terminal_conditions=["max_length", "ego_off_road"] (examples/ppo_agent.py:336), through the reference-shaped API."""
    import scenario_gym_amd as sga

    g = load_golden("roads")
    names = [str(n) for n in g["scenarios"]]
    nets = {}
    for n in names:
        k = str(g[f"{n}/network"])
        nets.setdefault(k, _road_network_from_golden(g, k))
    all_layers = [str(x) for x in g["layers"]]
    assert sga.RasterizedMapSensor._all_layers == all_layers
    for n in names:
        sc = scenario_from_arrays(scenario_arrays(g, f"{n}/scenario"), g[f"{n}/scenario/refs"])
        sc.road_network = nets[str(g[f"{n}/network"])]
        gym = sga.ScenarioGym(timestep=0.1)
        gym.set_scenario(sc)
        e = gym.state.scenario.entities[0]
        sensor = sga.RasterizedMapSensor(e, height=30, width=30, n=61)
        out = sensor.reset(gym.state).map
        want = g[f"{n}/map0"][0].astype(bool)  # [layer][61][61]
        assert out.shape == (61, 61, 2) and out[..., 1].any() and out[30, 30, 0]
        assert np.array_equal(out.transpose(2, 0, 1), want[:2])
        # the answer of a terminal condition does not depend on which one was asked first (flags cached before
        # the road network reached the device said "off the road"); out[30, 30, 0]: the ego stands on the driveable surface
        gym2 = sga.ScenarioGym(timestep=0.1)
        gym2.set_scenario(sc)
        sga.TERMINAL_CONDITIONS["collision"](gym2.state)  # (whatever it says: asked first, before the road network is on the device)
        assert not sga.TERMINAL_CONDITIONS["ego_off_road"](gym2.state)
        gym2.close()
        sensor = sga.RasterizedMapSensor(e, layers=all_layers, height=30, width=30, n=61, channels_first=True)
        sensor.reset(gym.state)
        for _ in range(30):
            gym.step()
        assert np.array_equal(sensor.step(gym.state).map, g[f"{n}/map0"][1].astype(bool))
        gym.close()
    # the terminal condition, four drifting egos as one batch
    scs = []
    for n in names:
        sc = scenario_from_arrays(scenario_arrays(g, f"{n}/drift/scenario"), g[f"{n}/drift/scenario/refs"])
        sc.road_network = nets[str(g[f"{n}/network"])]
        scs.append(sc)
    gym = sga.BatchedScenarioGym(timestep=0.1, terminal_conditions=["max_length", "ego_off_road"])
    gym.set_scenarios(scs)
    gym.rollout()
    for i, n in enumerate(names):
        assert gym.states[i].is_done and gym.states[i].t == g[f"{n}/drift/t"][-1], n
        assert bits_equal(gym.states[i].poses[scs[i].ego], g[f"{n}/drift/final_ego"]), n
    gym.close()


def test_device_resident_tick_equals_host_tick():
    """One RL tick with the policy's actions and the map observation staying on the GPU (torch tensor in, torch view out)
    gives the state and the map of the host-buffer calls."""
    import torch

    import scenario_gym_amd as sga
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    R, E = 48, 12
    packed = synthetic.make_batch(R, E, n_steps=200, ego_kind=L.KIND_AGENT_VEHICLE, extent=15.0)
    acts = synthetic.make_actions(30, R)
    sq = np.array([[-20.0, -20.0], [25.0, -20.0], [25.0, 18.0], [-20.0, 18.0]])
    net = dict(ring_off=[0, 1], vert_off=[0, 4], verts=sq, layers=[1])
    engines = []
    for _ in range(2):
        eng = sga.RolloutEngine(R, E)
        eng.upload(packed)
        eng.set_road_networks([net], np.zeros(R, np.int32))
        engines.append(eng)
    a, b = engines
    at = torch.as_tensor(acts, device="cuda:0")
    for k in range(30):
        a.step(1, acts[k:k + 1])
        b.step(1, at[k:k + 1])
        if k % 10 == 9:
            want = a.raster_map([0, 1], 30.0, 30.0, 24, 24)
            got = b.raster_map_torch([0, 1], 30.0, 30.0, 24, 24)
            assert got.is_cuda and got.dtype == torch.uint8 and tuple(got.shape) == (R, 2, 24, 24)
            assert np.array_equal(got.cpu().numpy().astype(bool), want) and want[:, 1].any() and not want[:, 1].all()
    sa, sb = a.state(), b.state()
    assert bits_equal(sa["poses"], sb["poses"]) and bits_equal(sa["vels"], sb["vels"])
    a.close()
    b.close()


def test_vector_env_episodes_match_oracle(oracle):
    """VectorScenarioEnv (the loop of integrations/openaigym.py:171-226 for a batch): seeded random policies drive the
    egos of the road-network scenarios until their episodes end (off the road / collision / max_length), environments
    restart one by one (sg_reset_scenarios) while the others run on.  Every episode of every environment -- its length,
    the ego pose after every tick, reward and done -- equals an oracle rollout of the same scenario with the same slice of
    actions; the observation is the map of the state it is returned with."""
    import scenario_gym_amd as sga
    import scenario_gym_amd._lib as L

    g = load_golden("roads")
    names = [str(n) for n in g["scenarios"]] * 2
    nets = {}
    scs, arrs = [], []
    for n in names:
        a = scenario_arrays(g, f"{n}/scenario")
        sc = scenario_from_arrays(a, g[f"{n}/scenario/refs"])
        k = str(g[f"{n}/network"])
        nets.setdefault(k, _road_network_from_golden(g, k))
        sc.road_network = nets[k]
        scs.append(sc)
        arrs.append(a)
    R, ticks = len(scs), 220
    env = sga.VectorScenarioEnv(scs, timestep=0.1, n=32)
    rng = np.random.default_rng(4)
    acts = np.stack([rng.uniform(-2, 3, (ticks, R)), rng.uniform(-0.5, 0.5, (ticks, R))], -1)
    obs = env.reset()
    assert obs.shape == (R, 2, 32, 32) and obs[:, 0, 16, 16].all() is not None
    start = np.zeros(R, int)   # tick at which the current episode of each env began
    episodes = 0
    ego_poses = [[env.engine.state()["poses"][r, arrs[r]["ego"]].copy()] for r in range(R)]
    for k in range(ticks):
        obs, reward, done, info = env.step(acts[k])
        st = env.engine.state()
        want_map = env.engine.raster_map([0, 1], 30.0, 30.0, 32, 32)
        assert np.array_equal(obs, want_map)
        for r in range(R):
            a = arrs[r]
            if not done[r]:
                ego_poses[r].append(st["poses"][r, a["ego"]].copy())
                assert reward[r] == 0.01
                continue
            # the episode that just ended: ticks start[r] .. k, against the oracle
            E = len(a["bbox"])
            kind = np.full(E, L.KIND_REPLAY, np.int32)
            kind[a["ego"]] = L.KIND_AGENT_VEHICLE
            ctrl = np.tile(oracle.DEFAULT_CTRL, (E, 1))
            ctrl[a["ego"], L.C_MAX_STEER], ctrl[a["ego"], L.C_MAX_ACCEL] = 0.9, 5.0
            n = k - start[r] + 1
            o = oracle.rollout(a["knot_off"], a["knots"], a["bbox"], a["etype"], kind, a["ego"], a["t0"], a["length"], 0.1,
                               terminal_mask=1 | 4 | 8, ctrl=ctrl, actions=np.vstack([acts[start[r]:k + 1, r], np.zeros((4, 2))]),
                               max_steps=n + 3, road=nets[str(g[f"{names[r]}/network"])]._arrays)
            assert o["n_steps"] == n and o["is_done"], (r, k, o["n_steps"], n)
            assert bits_equal(np.array(ego_poses[r]), o["poses"][:n, a["ego"]]), (r, k)
            hit = info["terminal_flags"][r] & (4 | 8)
            assert reward[r] == (-1.0 if hit else 0.01)
            # auto reset: the env is back at its start state
            assert st["n_steps"][r] == 0 and st["t"][r] == a["t0"] and not st["done"][r]
            ego_poses[r] = [st["poses"][r, a["ego"]].copy()]
            start[r] = k + 1
            episodes += 1
    env.close()
    assert episodes >= R  # every environment finished at least one episode on average


def test_graph_tick_on_scenarios_beyond_512_entities():
    """sg_tick on scenarios of 700 entities (the step is four kernels there, the observation the tiled entity raster + the
    terminal-condition kernel, all in one captured graph), with a road network under two of the three scenarios: same state,
    flags and maps as the three separate calls, through a restart of the scenarios that are done."""
    import scenario_gym_amd as sga
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    R, E = 3, 700
    packed = synthetic.make_batch(R, E, n_steps=60, ego_kind=L.KIND_AGENT_VEHICLE, extent=70.0, vanish_frac=0.2)
    packed.length = packed.length * np.array([0.3, 1.0, 1.0])
    acts = synthetic.make_actions(40, R)
    sq = np.array([[-9.0, -9.0], [12.0, -9.0], [12.0, 10.0], [-9.0, 10.0]])
    net = dict(ring_off=[0, 1], vert_off=[0, 4], verts=sq, layers=[1 | 2])
    a, b = (sga.RolloutEngine(R, E, terminal_conditions=["max_length", "ego_collision"]) for _ in range(2))
    for e in (a, b):
        e.upload(packed)
        e.set_road_networks([net], np.array([0, -1, 0], np.int32))
    geo = dict(layers=[0, 1], width=30.0, height=24.0, nw=20, nh=16)
    seen = 0
    for k in range(40):
        if k == 25:
            m = a.state()["done"]
            assert m.any()
            a.reset_scenarios(m)
            b.reset_scenarios(m)
        a.step(1, acts[k:k + 1])
        want_fl, want_map = a.terminal_flags(), a.raster_map(geo["layers"], geo["width"], geo["height"], geo["nw"], geo["nh"])
        obs, fl = b.tick(acts[k], **geo)
        assert np.array_equal(fl, want_fl) and np.array_equal(obs, want_map), k
        seen += int(obs[:, 0].sum())
    sa, sb = a.state(), b.state()
    for key in ("poses", "vels", "dists", "ctrl_state", "t", "n_steps", "done"):
        assert bits_equal(sa[key], sb[key]), key
    assert seen > 100 and not obs[1, 1].any()  # (scenario 1 has no road network: an empty surface)
    assert (fl[1] & 8) != 0  # ... and its entity 0 is off the road (flags are evaluated whatever the handle's terminal mask says)
    a.close()
    b.close()


def test_graph_tick_equals_separate_calls():
    """sg_tick (step + terminal flags + map observation replayed as one captured hipGraph) against the three separate calls,
    tick by tick: same state, same flags, same maps -- through restarts of single scenarios, a change of the time step, a
    change of the observation geometry and a new batch on the same handle (each rebuilds the graph)."""
    import torch

    import scenario_gym_amd as sga
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    R, E = 96, 10
    packed = synthetic.make_batch(R, E, n_steps=150, ego_kind=L.KIND_AGENT_VEHICLE, extent=14.0)
    acts = synthetic.make_actions(80, R)
    sq = np.array([[-9.0, -9.0], [12.0, -9.0], [12.0, 10.0], [-9.0, 10.0]])
    net = dict(ring_off=[0, 1], vert_off=[0, 4], verts=sq, layers=[1 | 2])
    a, b = sga.RolloutEngine(R, E, terminal_conditions=["max_length", "ego_collision", "ego_off_road"]), \
        sga.RolloutEngine(R, E, terminal_conditions=["max_length", "ego_collision", "ego_off_road"])
    for e in (a, b):
        e.upload(packed)
        e.set_road_networks([net], np.zeros(R, np.int32))
    geo = dict(layers=[0, 1], width=24.0, height=24.0, nw=16, nh=16)
    at = torch.as_tensor(acts, device="cuda:0")
    for k in range(60):
        if k == 20:  # restart the scenarios that are done, on both
            m = a.state()["done"]
            assert m.any()
            a.reset_scenarios(m)
            b.reset_scenarios(m)
        if k == 30:
            a.lib.sg_set_timestep(a.h, 0.05)
            b.lib.sg_set_timestep(b.h, 0.05)
        if k == 40:
            geo = dict(layers=[2, 0, 1], width=30.0, height=18.0, nw=20, nh=12)
        a.step(1, acts[k:k + 1])
        want_fl, want_map = a.terminal_flags(), a.raster_map(geo["layers"], geo["width"], geo["height"], geo["nw"], geo["nh"])
        if k % 2:
            obs, fl = b.tick(acts[k], **geo)
        else:
            obs, fl = b.tick(at[k], torch_out=True, **geo)
            obs, fl = obs.cpu().numpy().astype(bool), fl.cpu().numpy().astype(np.uint32)
        assert np.array_equal(fl, want_fl) and np.array_equal(obs, want_map), k
    sa, sb = a.state(), b.state()
    for key in ("poses", "vels", "dists", "ctrl_state", "t", "n_steps", "done"):
        assert bits_equal(sa[key], sb[key]), key
    # a new batch on the same handle
    packed2 = synthetic.make_batch(R, E, n_steps=150, ego_kind=L.KIND_AGENT_VEHICLE, extent=14.0, seed=77)
    for e in (a, b):
        e.upload(packed2)
        e.set_road_networks([net], np.zeros(R, np.int32))
    a.step(1, acts[0:1])
    obs, fl = b.tick(acts[0], **geo)
    assert np.array_equal(fl, a.terminal_flags()) and bits_equal(a.state()["poses"], b.state()["poses"])
    assert np.array_equal(obs, a.raster_map(geo["layers"], geo["width"], geo["height"], geo["nw"], geo["nh"]))
    a.close()
    b.close()


def test_pedestrian_agents_beside_a_building_through_the_gym():
    """PedestrianAgent + SocialForce on a scenario whose road network has a building (examples/crowds.py:149-205): the gym
    hands the network to the device by itself; the reference's closed loop is reproduced (<= 1e-8)."""
    import scenario_gym_amd as sga
    from scenario_gym_amd.road_network import RoadNetwork

    g = load_golden("ped_roads")
    rn = RoadNetwork(name="crowds")
    rn._arrays = {k: g[f"net/{k}"] for k in ("ring_off", "vert_off", "verts", "layers")}
    si, dt, p = 0, 0.1, "loop0/dt10"
    sc = scenario_from_arrays(scenario_arrays(g, f"loop{si}/scenario"), g[f"loop{si}/scenario/refs"])
    sc.road_network = rn
    routes, vdes, thr = g[f"loop{si}/routes"], g[f"loop{si}/vdes"], float(g[f"loop{si}/distance_threshold"])
    idx = {e.ref: k for k, e in enumerate(sc.entities)}

    def create_agent(s, e):
        if e.ref == "ego":
            return sga.ReplayTrajectoryAgent(e)
        return sga.PedestrianAgent(e, routes[idx[e.ref]], vdes[idx[e.ref]],
                                   sga.SocialForce(sga.SocialForceParameters(std_lon=0.0, std_lat=0.0)), distance_threshold=thr)

    gym = sga.ScenarioGym(timestep=dt)
    gym.set_scenario(sc, create_agent=create_agent)
    gym.rollout()
    ref = g[p + "/poses"][-1]
    got = np.array([gym.state.poses[e] for e in sc.entities])
    assert gym.state.t == g[p + "/t"][-1] and np.abs(got - ref).max() < 1e-8
    forces = np.array([gym.state.agents[e].force for e in sc.entities[1:]])
    assert np.abs(forces - g[p + "/extra"][-1][1:, 2:]).max() < 1e-8
    gym.close()


def test_collision_metric_types_through_the_gym(oracle):
    """CollisionMetric through ScenarioGym on the hand-made scenes of collision_types.npz: the names the reference's code
    gave (head_on, rear_end, t_bone, side_swipe, non_vehicle); c_tol reaches the device (a huge tolerance turns every
    contact point into a corner)."""
    import scenario_gym_amd as sga

    g = load_golden("collision_types")
    types = [str(t) for t in g["types"]]
    for n in ("head_on", "rear_end", "rear_ended", "t_bone", "t_boned", "side_swipe", "oblique", "rand3", "rand27"):
        sc = scenario_from_arrays(scenario_arrays(g, f"{n}/scenario"), g[f"{n}/scenario/refs"])
        gym = sga.ScenarioGym(timestep=0.05, metrics=[sga.CollisionMetric(), sga.CollisionPointMetric()])
        gym.set_scenario(sc)
        gym.rollout()
        got = gym.get_metrics()["collisions"]
        pts = gym.get_metrics()["collision_points"]
        assert [r for r, _, _ in pts] == [r for _, r, _ in got]
        assert all(np.abs(np.array([*p, a]) - w).max() < 1e-10 for (_, p, a), w in zip(pts, g[f"{n}/ev_point"]))
        want = [(float(t), str(g[f"{n}/scenario/refs"][o]), types[k]) for t, o, k in zip(g[f"{n}/ev_t"], g[f"{n}/ev_other"], g[f"{n}/ev_type"])]
        assert got == want, (n, got, want)
        gym.close()
    sc = scenario_from_arrays(scenario_arrays(g, "rear_end/scenario"), g["rear_end/scenario/refs"])
    gym = sga.ScenarioGym(timestep=0.05, metrics=[sga.CollisionMetric(c_tol=3.0)])
    gym.set_scenario(sc)
    gym.rollout()
    # every contact angle now falls into a front-corner window of both boxes: front/front at collision angle 0 is a side swipe
    assert [c for _, _, c in gym.get_metrics()["collisions"]] == ["side_swipe"]
    gym.close()


def test_rss_metric_through_the_gym():
    """tests/test_rss.py:5-25: ScenarioGym(state_callbacks=[RSSDistances()], metrics=[RSS()]) -- the metric dictionary has
    the two boolean RSS entries; their values, and the callback's safe distances along the way, equal the reference's."""
    import scenario_gym_amd as sga

    g = load_golden("rss")
    for n in ("a5e43fe4-646a-49ba-82ce-5f0063776566", "e1bdb607-206b-4f40-9bc4-59ded182ecc8", "synth3", "synth9"):
        sc = scenario_from_arrays(scenario_arrays(g, f"{n}/scenario"), g[f"{n}/scenario/refs"])
        cb = sga.RSSDistances()
        gym = sga.ScenarioGym(timestep=0.1, state_callbacks=[cb], metrics=[sga.RSS()])
        gym.set_scenario(sc)
        k = 0
        while not gym.state.is_done:
            gym.step()
            k += 1
            if k % 25 == 0:
                want_code, want_safe = g[f"{n}/code"][k], g[f"{n}/safe"][k]
                sd = cb.safe_distances(gym.state)
                recs = cb.latest_records(gym.state)
                for j, e in enumerate(sc.entities):
                    assert (e in sd) == (want_code[j] >= 0)
                    if e in sd:
                        assert np.abs(np.array(sd[e]) - want_safe[j]).max() < 1e-9 and recs[e] == cb.CODES[want_code[j]]
        data = gym.get_metrics()
        assert type(data["RSS_safe_longitudinal"]) is bool and type(data["RSS_safe_lateral"]) is bool
        assert data["RSS_safe_longitudinal"] == bool(g[f"{n}/safe_longitudinal"]), n
        assert data["RSS_safe_lateral"] == bool(g[f"{n}/safe_lateral"]), n
        gym.close()
    with pytest.raises(ValueError):
        gym = sga.ScenarioGym(timestep=0.1, metrics=[sga.RSS()])
        gym.set_scenario(sc)
        gym.rollout()
        gym.get_metrics()
    # the whole file list as one batch, one rollout() call (the callback does not force the per-step host path)
    names = [str(x) for x in g["names"]]
    scs = [scenario_from_arrays(scenario_arrays(g, f"{n}/scenario"), g[f"{n}/scenario/refs"]) for n in names]
    gym = sga.BatchedScenarioGym(timestep=0.1, state_callbacks=[sga.RSSDistances()], metrics=lambda: [sga.RSS(), sga.EgoMaxSpeed()])
    gym.set_scenarios(scs)
    assert not gym._per_step_host_path()
    gym.rollout()
    for n, m in zip(names, gym.get_metrics()):
        assert m["RSS_safe_longitudinal"] == bool(g[f"{n}/safe_longitudinal"]) and m["RSS_safe_lateral"] == bool(g[f"{n}/safe_lateral"]), n
    gym.close()


def test_bench_two_live_ranks_on_one_gpu(tmp_path):
    """`python bench.py --gpus 2` with the REAL engine: bench.py starts its two ranks itself; both share GPU 0 here
    (SGYM_DIST_ONE_DEVICE; gloo carries the dispatch / collection bytes because RCCL refuses two ranks on one device).  Rank 0's
    line has both ranks; `value` is the batch SPLIT over the ranks (scaling "strong": BASELINE.json's metric as written), the weak
    and the strong_sliced blocks ride beside it, and every block's scenarios are verified against the oracle."""
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(SGYM_DIST_BACKEND="gloo", SGYM_DIST_ONE_DEVICE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scenarios", "1024", "--sim-steps", "2000",
                          "--steps", "2", "--warmup", "1", "--verify", "4"], capture_output=True, text=True, timeout=600, env=env,
                         cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["ranks"] == 2 and line["n_gpus"] == 2 and line["backend"] == "gloo" and line["scaling"] == "strong"
    assert line["config"]["scenarios_per_gpu"] == 1024 // 2 and line["weak"]["scenarios_per_gpu"] == 1024
    assert line["strong_sliced"]["scenarios_per_gpu"] == 512 and "configs" not in line
    for blk in (line, line["weak"], line["strong_sliced"]):
        assert blk["verified"]["equal"] and blk["verified"]["scenarios"] >= 4 and len(blk["per_rank_value"]) == 2
    assert line["value"] > 0 and line["strong_sliced"]["value"] > 0


def test_rccl_initialised_first_then_the_persistent_launch(tmp_path):
    """Multi-GPU readiness that one GPU can prove: with the REAL `nccl` backend initialised first (SGYM_FORCE_DIST=1: one rank;
    the dispatch broadcast and the metric gather go through RCCL, whose streams and proxy thread exist before the engine
    does) the 4096 x 64 batch still runs as the one persistent launch, and the bench line says so (`roofline.schedule`, no
    "degraded" key, `--require-queue` passes).  Rounds 3-4 needed a timing probe here (how many of the engine's streams the
    runtime overlapped); there is nothing left to probe.  Two `python bench.py` subprocesses, both oracle-verified; their
    throughputs are reported, not compared (schedule assertions only: VERDICT r4, weak 8)."""
    import json
    import subprocess
    import sys

    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "SGYM_FORCE_DIST")}

    def run(extra):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--verify", "4",
                              "--no-cpu-baseline", "--require-queue", "--no-configs"], capture_output=True, text=True, timeout=900, env=dict(base, **extra),
                             cwd=str(tmp_path))
        assert out.returncode == 0, out.stderr[-3000:]
        return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])

    plain, rccl = run({}), run({"SGYM_FORCE_DIST": "1", "MASTER_PORT": "29533"})
    assert plain["backend"] == "none" and rccl["backend"] == "nccl" and rccl["ranks"] == 1
    for ln in (plain, rccl):
        sc = ln["roofline"]["schedule"]
        assert sc["per_rank"] == ["persistent_queue"] and sc["launches_per_rollout"] == 1 and sc["blocks_per_rank"] == 4096, sc
        assert sc["table_ring"] == sc["chunks"] and sc["wavefronts"] == 3 * sc["simds"], sc
        assert "degraded" not in ln and ln["verified"]["equal"] and ln["engine"].startswith("scenario_gym_amd.RolloutEngine")
        assert ln["roofline"]["bound"] == "valu_fp64" and 0.0 < ln["roofline"]["frac"] <= 1.0
    print(f"persistent launch: {plain['value'] / 1e9:.1f} G alone, {rccl['value'] / 1e9:.1f} G with RCCL initialised first")


def test_library_names_the_kernel_it_launched(monkeypatch):
    """roofline.kernel in bench.py's line is what the library says it launched (sg_last_kernel), not what the shape implies:
    the persistent table launch and its chunk-launch fallback, the crowd kernel and the general pedestrian kernel it falls
    back to, the time-sliced replay."""
    import scenario_gym_amd as sga
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    def name(packed, steps, **env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = sga.RolloutEngine(packed.n_scenarios, packed.n_entities)
        eng.upload(packed)
        eng.rollout(steps)
        n = eng.last_kernel()
        eng.close()
        for k in env:
            monkeypatch.delenv(k)
        return n

    pid = synthetic.make_batch(128, 64, n_steps=300, ego_kind=L.KIND_AGENT_PID)
    assert name(pid, 300) == "sg::rollout_kernel_tabq_planar<64>"
    assert name(pid, 300, SG_QUEUE="0") == "sg::rollout_kernel_tab_planar<64>"
    assert name(pid, 300, SG_PLANAR="0") == "sg::rollout_kernel_tabq<64>"
    assert name(pid, 4) == "sg::rollout_kernel<64, 1, false, false>"  # (a short call keeps its controllers in the kernel)
    crowd = synthetic.make_crowd(8, 200, n_steps=100)
    assert name(crowd, 100) == "sg::rollout_kernel_crowd<4>"
    assert name(crowd, 100, SG_CROWD_KERNEL="0") == "sg::rollout_kernel<64, 4, true, false>"
    replay = synthetic.make_batch(64, 16, n_steps=400, ego_kind=L.KIND_AGENT_REPLAY)
    assert name(replay, 400) in ("sg::rollout_kernel_slice<16>", "sg::rollout_kernel<16, 1, false, true>")


def test_bench_line_carries_every_single_gpu_config(tmp_path):
    """The driver times ONE `python bench.py` line: besides the c3 headline it carries the other single-GPU BASELINE configs -- c2
    (256 x 16 replay, the state of every step materialised) and c5 (1024 x 256 crowd) -- each with its own timed passes, roofline
    fraction and oracle verification; the headline's own keys are what they were."""
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--verify", "2", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["scaling"] == "weak" and line["n_gpus"] == 1 and line["config"]["scenarios_per_gpu"] == 4096 and line["verified"]["equal"]
    cf = line["configs"]
    assert set(cf) == {"c2", "c5"}
    assert (cf["c2"]["scenarios"], cf["c2"]["entities"], cf["c2"]["sim_steps"]) == (256, 16, 10000)
    assert (cf["c5"]["scenarios"], cf["c5"]["entities"], cf["c5"]["sim_steps"]) == (1024, 256, 10000)
    for c in cf.values():
        assert c["value"] > 0 and c["steps"] == 3 and c["verified"]["equal"] and c["verified"]["steps"] == 10000
        assert 0.0 < c["roofline"]["frac"] <= 1.0 and c["roofline"]["kernel_ms"] > 0 and c["roofline"]["kernel"].startswith("sg::rollout_kernel")
    print(f"c3 {line['value'] / 1e9:.1f} G, c2 {cf['c2']['value'] / 1e9:.2f} G, c5 {cf['c5']['value'] / 1e9:.2f} G")


def test_chunk_launch_fallback_is_flagged(tmp_path):
    """SG_QUEUE=0 runs the table path as the chunk launches of rounds 1-4: the line says so (stderr + a "degraded" key),
    `--require-queue` turns that into exit code 4.  Results do not depend on the schedule (verified)."""
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["SG_QUEUE"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--scenarios", "1024", "--sim-steps", "1000", "--steps", "2", "--warmup", "1",
           "--verify", "4", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    sc = line["roofline"]["schedule"]
    assert sc["per_rank"] == ["chunk_launches"] and sc.get("degraded") and sc["launches_per_rollout"] > 1
    assert "degraded" in line and "DEGRADED" in out.stderr and line["verified"]["equal"]
    out = subprocess.run(cmd + ["--require-queue"], capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert out.returncode == 4, (out.returncode, out.stderr[-2000:])
