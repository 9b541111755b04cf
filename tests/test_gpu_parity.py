"""GPU parity: the HIP engine (through the C ABI) against golden vectors from the real reference
and against the CPU oracle on seeded synthetic batches.  Run on the MI355X box with -m gpu.

Bars: replay poses, t, step counts, velocities, distances, ego metrics: bit-identical;
controller-integrated poses: <= 1e-5 abs (observed ~1e-10 vs the reference, 0 vs the oracle);
collision adjacency and events: exact.
"""
import os
import sys

import numpy as np
import pytest

from conftest import bits_equal, load_golden, scenario_arrays

pytestmark = pytest.mark.gpu

CTRL_TOL = 1e-9


@pytest.fixture(scope="module")
def sga():
    import scenario_gym_amd as sga
    import scenario_gym_amd._lib as L

    L.load()
    return sga


def _engine_run(sga, packed, dt, n_max, persist=False, terminal=None, actions=None, ev_cap=64, tuning=None):
    """reset + rollout with full recording; returns per-scenario results shaped like the goldens."""
    eng = sga.RolloutEngine(packed.n_scenarios, packed.n_entities, timestep=dt, persist=persist,
                            terminal_conditions=terminal, record_capacity=n_max + 1, event_capacity=ev_cap)
    if tuning is not None:
        eng.set_tuning(**tuning)
    eng.upload(packed)
    if actions is None:
        eng.rollout(n_max)
    else:  # external actions: gym.step() per action row, scenario by scenario stops being compared at done
        eng.step(actions.shape[0], actions)
    st = eng.state()
    rows, events = eng.metrics()
    t, poses = eng.record(n_max + 1)
    eng.close()
    return st, rows, events, t, poses


def _dense(coll_row, E):
    return ((coll_row[:, None] >> np.arange(E, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.uint8)


def _compare_final(g, p, r, E, st, rows, events, t, poses, exact, n_rec=None):
    n = int(g[p + "/n_steps"])
    assert rows["n_steps"][r] == n, (p, rows["n_steps"][r], n)
    assert bits_equal(t[: n + 1, r], g[p + "/t"]), p
    got = poses[: n + 1, r, :E]
    ref = g[p + "/poses"]
    if exact:
        assert bits_equal(got, ref), p
    else:
        assert np.array_equal(np.isnan(got), np.isnan(ref)), p
        assert np.nanmax(np.abs(got - ref)) < CTRL_TOL, p
    for k, key in (("vels", "vels"), ("dists", "dists")):
        a, b = st[k][r, :E], g[p + "/" + key][-1]
        if exact:
            assert bits_equal(a, b), (p, k)
        else:
            assert np.array_equal(np.isnan(a), np.isnan(b)) and np.nanmax(np.abs(a - b)) < CTRL_TOL, (p, k)
    assert np.array_equal(_dense(st["coll"][r, :E], E), g[p + "/coll"][-1]), p
    for k in ("ego_avg_speed", "ego_max_speed", "ego_distance_travelled"):
        a, b = rows[k][r], float(g[p + "/metric_" + k])
        assert (a == b) if exact else (abs(a - b) < CTRL_TOL), (p, k, a, b)
    if p + "/ev_t" in g:
        ev = events[events["scenario"] == r]
        assert np.array_equal(ev["t"], g[p + "/ev_t"]) and np.array_equal(ev["other"], g[p + "/ev_other"]), p


def test_xosc_scenarios_bit_identical(sga):
    from scenario_gym_amd.packing import pack_arrays

    g = load_golden("scenarios")
    names = list(g["names"])
    scs = [scenario_arrays(g, f"{n}/scenario") for n in names]
    packed = pack_arrays(scs)
    for run, dt in (("dt30", 1 / 30), ("dt10", 0.1)):
        nmax = max(int(g[f"{n}/{run}/n_steps"]) for n in names) + 2
        st, rows, events, t, poses = _engine_run(sga, packed, dt, nmax)
        for r, n in enumerate(names):
            _compare_final(g, f"{n}/{run}", r, len(scs[r]["etype"]), st, rows, events, t, poses, True)


def test_vanishing_and_persist(sga):
    from scenario_gym_amd.packing import pack_arrays

    g = load_golden("scenarios")
    sc = scenario_arrays(g, "vanish/scenario")
    packed = pack_arrays([sc])
    for run, persist in (("nopersist", False), ("persist", True)):
        out = _engine_run(sga, packed, 0.1, int(g[f"vanish/{run}/n_steps"]) + 2, persist=persist)
        _compare_final(g, f"vanish/{run}", 0, len(sc["etype"]), *out, True)
        if persist:
            assert not np.isnan(out[4][: int(g["vanish/persist/n_steps"]) + 1, 0]).any()


@pytest.mark.parametrize("dtn,dt", [("dt30", 1 / 30), ("dt10", 0.1)])
def test_synthetic_goldens(sga, dtn, dt):
    """4 small scenes through the reference: replay (persist on/off), collision terminals, PID ego,
    external-action VehicleController ego -- all four scenes in one batch per configuration."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd.packing import default_kinds, pack_arrays

    g = load_golden("synth")
    n = int(g["n"])
    scs = [scenario_arrays(g, f"{i}/scenario") for i in range(n)]
    Es = [len(s["etype"]) for s in scs]

    def run(tag, exact, kinds=None, **kw):
        nmax = max(int(g[f"{i}/{tag}/n_steps"]) for i in range(n)) + 2
        out = _engine_run(sga, pack_arrays(scs, kinds=kinds), dt, nmax, **kw)
        for i in range(n):
            _compare_final(g, f"{i}/{tag}", i, Es[i], *out, exact)

    run(f"replay_{dtn}_nopersist", True)
    run(f"replay_{dtn}_persist", True, persist=True)
    run(f"term_collision_{dtn}", True, terminal=["max_length", "collision"])
    run(f"term_ego_collision_{dtn}", True, terminal=["max_length", "ego_collision"])
    kinds = []
    for s, E in zip(scs, Es):
        k = default_kinds(E, s["ego"])
        k[s["ego"]] = L.KIND_AGENT_PID
        kinds.append(k)
    run(f"pid_{dtn}", False, kinds=kinds)
    # external actions: every scene has its own action sequence; gym.step() n times
    for k in kinds:
        k[k == L.KIND_AGENT_PID] = L.KIND_AGENT_VEHICLE
    tag = f"ext_{dtn}"
    steps = [int(g[f"{i}/{tag}/n_steps"]) for i in range(n)]
    assert len(set(steps)) == 1  # same length L and dt => same step count
    acts = np.stack([g[f"{i}/{tag}/actions"][: steps[0]] for i in range(n)], axis=1)
    out = _engine_run(sga, pack_arrays(scs, kinds=kinds), dt, steps[0], actions=acts)
    for i in range(n):
        _compare_final(g, f"{i}/{tag}", i, Es[i], *out, False)


def test_pid_agent_on_xosc(sga):
    """tests/test_controller.py:7-25 configuration (accel_Kp=2, max_accel=5, max_steer=pi/90)."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd.engine import DEFAULT_CTRL
    from scenario_gym_amd.packing import default_kinds, pack_arrays

    g = load_golden("pid_xosc")
    sc = scenario_arrays(g, "scenario")
    E = len(sc["etype"])
    kind = default_kinds(E, sc["ego"])
    kind[sc["ego"]] = L.KIND_AGENT_PID
    ctrl = np.tile(DEFAULT_CTRL, (E, 1))
    ctrl[:, L.C_ACCEL_KP], ctrl[:, L.C_MAX_ACCEL], ctrl[:, L.C_MAX_STEER] = g["params"]
    out = _engine_run(sga, pack_arrays([sc], kinds=[kind], ctrls=[ctrl]), 0.1, 230)
    _compare_final(g, "run", 0, E, *out, False)
    assert out[1]["n_steps"][0] == 225


def test_head_on_collision_event(sga):
    """tests/test_utils.py:12-61 scene at dt=0.1: one ego event at t=8.799999999999985, non_vehicle."""
    from scenario_gym_amd.packing import pack_arrays

    g = load_golden("collision")
    sc = scenario_arrays(g, "headon/scenario")
    out = _engine_run(sga, pack_arrays([sc]), 0.1, 110)
    _compare_final(g, "headon/run", 0, 2, *out, True)
    ev = out[2]
    assert ev["t"].tolist() == [8.799999999999985] and ev["other"].tolist() == [1] and ev["type"].tolist() == [5]


def test_stepwise_state_matches_reference(sga):
    """gym.step() one at a time: velocities, distances and collision rows after EVERY step."""
    from scenario_gym_amd.packing import pack_arrays

    g = load_golden("synth")
    scs = [scenario_arrays(g, f"{i}/scenario") for i in range(int(g["n"]))]
    packed = pack_arrays(scs)
    eng = sga.RolloutEngine(packed.n_scenarios, packed.n_entities, timestep=0.1)
    eng.upload(packed)
    n = int(g["0/replay_dt10_nopersist/n_steps"])
    for k in range(n + 1):
        st = eng.state()
        for i, sc in enumerate(scs):
            E = len(sc["etype"])
            p = f"{i}/replay_dt10_nopersist"
            assert st["t"][i] == g[p + "/t"][k]
            assert bits_equal(st["poses"][i, :E], g[p + "/poses"][k]), (i, k)
            assert bits_equal(st["vels"][i, :E], g[p + "/vels"][k]), (i, k)
            assert bits_equal(st["dists"][i, :E], g[p + "/dists"][k]), (i, k)
            assert np.array_equal(_dense(st["coll"][i, :E], E), g[p + "/coll"][k]), (i, k)
        if k < n:
            eng.step(1)
    assert eng.state()["done"].all()
    eng.close()


def test_timestep_change_mid_rollout(sga):
    """tests/test_scenario_gym.py:28-44: dt follows gym.timestep when it is changed between steps."""
    from scenario_gym_amd.packing import pack_arrays

    g = load_golden("scenarios")
    packed = pack_arrays([scenario_arrays(g, "a5e43fe4/scenario")])
    eng = sga.RolloutEngine(1, packed.n_entities, timestep=0.5)
    eng.upload(packed)
    eng.step(1)
    st = eng.state()
    assert np.allclose(st["t"] - st["prev_t"], 0.5)
    eng.set_timestep(0.2)
    eng.step(1)
    st = eng.state()
    assert np.allclose(st["t"], st["prev_t"] + 0.2)
    eng.close()


# --------------------------------------------------------------------------- oracle parity at size
def _oracle_batch(oracle, packed, dt, n_max, idxs, **kw):
    from scenario_gym_amd.packing import unpack_scenario

    out = {}
    for r in idxs:
        s = unpack_scenario(packed, r)
        out[r] = oracle.rollout(s["knot_off"], s["knots"], s["bbox"], s["etype"], s["kind"], s["ego"],
                                s["t0"], s["length"], dt, ctrl=s["ctrl"], max_steps=n_max, **kw)
    return out


@pytest.mark.parametrize("R,E,steps,ego_kind", [
    (256, 16, 400, "replay"),   # BASELINE config 2 shape
    (128, 64, 300, "pid"),      # BASELINE config 3 shape (PID ego)
    (64, 64, 200, "vehicle"),   # config 3 variant with external actions
    (96, 5, 150, "replay"),     # ragged width -> tile of 8 lanes
    (40, 33, 120, "pid"),       # 33 entities -> tile of 64 lanes with padding
])
def test_synthetic_batch_matches_oracle(sga, oracle, R, E, steps, ego_kind):
    """Every scenario of a seeded synthetic batch, all steps: poses bit-identical (controllers too:
    oracle and kernel share the same fp64 operation order), final state, metrics and events."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    kind = dict(replay=L.KIND_AGENT_REPLAY, pid=L.KIND_AGENT_PID, vehicle=L.KIND_AGENT_VEHICLE)[ego_kind]
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=kind, static_frac=0.15, vanish_frac=0.2, extent=40.0)
    acts = synthetic.make_actions(steps, R) if ego_kind == "vehicle" else None
    st, rows, events, t, poses = _engine_run(sga, packed, 1 / 30, steps, actions=acts, ev_cap=128)
    n_checked_events = 0
    for r in range(R):
        kw = dict(actions=acts[:, r], force_steps=True) if acts is not None else {}
        o = _oracle_batch(oracle, packed, 1 / 30, steps, [r], **kw)[r]
        n = o["n_steps"]
        assert rows["n_steps"][r] == n and rows["final_t"][r] == o["final_t"], r
        assert bits_equal(t[: n + 1, r], o["t"]), r
        assert bits_equal(poses[: n + 1, r], o["poses"]), r
        assert bits_equal(st["vels"][r], o["vels"][-1]) and bits_equal(st["dists"][r], o["dists"][-1]), r
        assert np.array_equal(st["coll"][r], o["coll"][-1, :, 0]), r
        for k in ("ego_avg_speed", "ego_max_speed", "ego_distance_travelled"):
            assert rows[k][r] == o["metric_" + k], (r, k)
        ev = events[events["scenario"] == r]
        assert rows["n_collisions"][r] == o["n_events"]
        assert np.array_equal(ev["t"], o["ev_t"]) and np.array_equal(ev["other"], o["ev_other"]), r
        n_checked_events += len(ev)
    assert n_checked_events > 0  # the dense scenes do collide


def test_identical_boxes_alias_rule(sga, oracle):
    """Two co-located static entities with bit-identical boxes never list each other and a third
    party sees only the LAST of them (state/utils.py:32-40, utils.py:59), twice in the metric."""
    from scenario_gym_amd.packing import pack_arrays

    z = [0.0] * 4
    knots = np.array([
        [0.0, -10.0, 0.0, *z], [10.0, 10.0, 0.0, *z],   # ego drives through the origin
        [0.0, 0.0, 0.0, *z],                            # static A
        [0.0, 0.0, 0.0, *z],                            # static B == A
        [0.0, 30.0, 30.0, *z],                          # far away
    ])
    sc = dict(knot_off=np.array([0, 2, 3, 4, 5]), knots=knots, bbox=np.tile([2.0, 4.0, 0.0, 0.0], (4, 1)),
              etype=np.full(4, 2, np.int32), ego=0, t0=0.0, length=10.0)
    st, rows, events, t, poses = _engine_run(sga, pack_arrays([sc]), 0.1, 110)
    from scenario_gym_amd.packing import default_kinds

    o = oracle.rollout(sc["knot_off"], knots, sc["bbox"], sc["etype"], default_kinds(4, 0), 0, 0.0, 10.0, 0.1)
    assert o["n_events"] == 2 and o["ev_other"].tolist() == [2, 2]
    assert events["other"].tolist() == [2, 2] and np.array_equal(events["t"], o["ev_t"])
    assert rows["n_collisions"][0] == 2
    assert np.array_equal(st["coll"][0], o["coll"][-1, :, 0])


@pytest.mark.parametrize("E", [300, 700])
def test_identical_boxes_in_wide_scenarios(sga, oracle, E):
    """The alias rule (state/utils.py:32-40, utils.py:59) in scenarios of several row words and beyond 512 entities, where the
    collision rows are built by wide_collide_kernel and the twins' bits are moved by wide_owner_kernel: groups of two to four
    entities with bit-identical trajectories and boxes -- far apart in the entity order, across word and workgroup
    boundaries -- never list each other, everybody else lists the LAST of a group; rows of every step's final state,
    events of the ego (which drives through a group) and their order equal the oracle's."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic
    from scenario_gym_amd.packing import unpack_scenario

    R, steps = 3, 50
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=L.KIND_AGENT_REPLAY, static_frac=0.2, vanish_frac=0.1, extent=45.0, n_knots=8)
    rng = np.random.default_rng(E)
    off = packed.knot_off
    rows = [packed.knots[off[i]:off[i + 1]].copy() for i in range(R * E)]
    n_groups = 0
    for r in range(R):
        for _ in range(12):  # copy entity a's knots and box onto 1..3 others of the same scenario
            a = int(rng.integers(1, E))
            for b in rng.choice(np.arange(1, E), int(rng.integers(1, 4)), replace=False):
                rows[r * E + int(b)] = rows[r * E + a].copy()
                packed.bbox[r * E + int(b)] = packed.bbox[r * E + a]
            n_groups += 1
        # ... and one group sits on the ego's path: the ego meets twins
        ego = rows[r * E]
        mid = ego[len(ego) // 2].copy()
        for b in (E // 3, E - 2):
            rows[r * E + b] = mid[None, :].copy()
            packed.bbox[r * E + b] = packed.bbox[r * E + E // 3]
    packed.knots = np.concatenate(rows)
    packed.knot_off = np.concatenate([[0], np.cumsum([len(x) for x in rows])]).astype(np.int64)
    packed = packed.validate()
    eng = sga.RolloutEngine(R, E, event_capacity=256)
    eng.upload(packed)
    eng.rollout(steps)
    st = eng.state()
    mrows, events = eng.metrics()
    eng.close()
    met = 0
    for r in range(R):
        s = unpack_scenario(packed, r)
        o = oracle.rollout(s["knot_off"], s["knots"], s["bbox"], s["etype"], s["kind"], s["ego"], s["t0"], s["length"], 1 / 30,
                           ctrl=s["ctrl"], max_steps=steps, event_cap=256)
        assert mrows["n_steps"][r] == o["n_steps"]
        assert np.array_equal(_dense_words(st["coll"][r], E), oracle.coll_to_dense(o["coll"], E)[-1]), r
        ev = events[events["scenario"] == r]
        assert mrows["n_collisions"][r] == o["n_events"] and np.array_equal(ev["t"], o["ev_t"]) and np.array_equal(ev["other"], o["ev_other"]), r
        met += int((ev["other"] == E - 2).sum())
        assert not (ev["other"] == E // 3).any()  # the earlier twin is never the one that is listed
    assert met >= R and n_groups == 12 * R


def test_full_size_invariants(sga, oracle):
    """BASELINE config 3 width (4096 x 64, PID ego) for 300 steps: size-independent properties +
    oracle spot checks on scattered scenarios."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    R, E, steps = 4096, 64, 300
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=L.KIND_AGENT_PID)
    eng = sga.RolloutEngine(R, E, event_capacity=32)
    eng.upload(packed)
    eng.rollout(steps)
    st = eng.state()
    rows, events = eng.metrics()
    assert (rows["n_steps"] == steps).all() and rows["done"].all()
    # one shared clock: t accumulates t += dt in fp64 identically in every scenario
    tt = 0.0
    for _ in range(steps):
        tt += 1 / 30
    assert (rows["final_t"] == tt).all()
    dense = ((st["coll"][:, :, None] >> np.arange(E, dtype=np.uint64)) & np.uint64(1)).astype(bool)
    assert np.array_equal(dense, dense.transpose(0, 2, 1))          # adjacency is symmetric
    assert not dense[:, np.arange(E), np.arange(E)].any()            # nobody collides with itself
    assert not dense[~st["present"]].any()                           # absent entities have empty rows
    assert (st["dists"] >= 0).all() and (rows["ego_max_speed"] >= rows["ego_avg_speed"] * 0).all()
    # idempotence: a second rollout from reset reproduces the same bits
    eng.rollout(steps)
    st2 = eng.state()
    for k in ("poses", "vels", "dists", "coll"):
        assert bits_equal(st[k].astype(np.float64), st2[k].astype(np.float64)), k
    eng.close()
    for r in (0, 1, 777, 2048, 4095):
        o = _oracle_batch(oracle, packed, 1 / 30, steps, [r])[r]
        assert bits_equal(st["poses"][r], o["poses"][-1]) and bits_equal(st["vels"][r], o["vels"][-1])
        assert np.array_equal(st["coll"][r], o["coll"][-1, :, 0])
        assert rows["ego_distance_travelled"][r] == o["metric_ego_distance_travelled"]


def test_full_size_crowd_invariants(sga, oracle):
    """BASELINE config 5 at its full width (1024 scenarios x 256 pedestrians, the crowd kernel) for 300 steps, through the
    densest phase: size-independent properties of the state + oracle spot checks on scattered scenarios."""
    from scenario_gym_amd import synthetic

    R, E, steps = 1024, 256, 300
    packed = synthetic.make_crowd(R, E, n_steps=steps)
    eng = sga.RolloutEngine(R, E, event_capacity=64)
    eng.upload(packed)
    eng.rollout(steps)
    st = eng.state()
    rows, events = eng.metrics()
    assert (rows["n_steps"] == steps).all() and rows["done"].all() and st["present"].all()
    tt = 0.0
    for _ in range(steps):
        tt += 1 / 30
    assert (rows["final_t"] == tt).all()
    bits = np.unpackbits(np.ascontiguousarray(st["coll"]).view(np.uint8), axis=-1, bitorder="little").reshape(R, E, -1)[:, :, :E].astype(bool)
    assert np.array_equal(bits, bits.transpose(0, 2, 1)) and not bits[:, np.arange(E), np.arange(E)].any()
    assert bits.any(axis=(1, 2)).mean() > 0.9                      # almost every crowd has colliding boxes by now
    # overlapping boxes have centres within one box diagonal; pedestrians further apart than that never collide
    xy = st["poses"][:, :, :2]
    d = np.linalg.norm(xy[:, :, None, :] - xy[:, None, :, :], axis=-1)
    diag = float(np.hypot(*synthetic.PEDESTRIAN1_BBOX[:2]))
    assert (d[bits] <= diag + 1e-12).all()
    assert not bits[d > diag + 1e-12].any()
    goal = st["ctrl_state"][:, :, 1]
    arrived = goal > 1
    assert set(np.unique(goal)) <= {1.0, 2.0} and 0 < arrived.mean() < 0.5
    assert (st["force"][arrived] == 0).all() and (st["ctrl_state"][:, :, 0][arrived] == 0).all()   # agent.py:65-68
    vdes = packed.ctrl[:, 9].reshape(R, E)
    assert (st["ctrl_state"][:, :, 0] <= vdes * 1.3 + 1e-12).all()  # speed = min(|F|, speed_desired * max_speed_factor)
    assert (st["dists"] >= 0).all() and (st["dists"][~arrived] > 0).all()
    ev_r = np.bincount(events["scenario"], minlength=R)
    assert (rows["n_collisions"] >= ev_r).all() and (events["type"] == 5).all()   # pedestrians: "non_vehicle"
    # idempotence: a second rollout from reset reproduces the same bits
    eng.rollout(steps)
    st2 = eng.state()
    for k in ("poses", "vels", "dists", "force"):
        assert bits_equal(st[k], st2[k]), k
    assert np.array_equal(st["coll"], st2["coll"])
    eng.close()
    for r in (0, 511, 1023):
        o = _oracle_one(oracle, packed, r, 1 / 30, steps)
        assert bits_equal(st["poses"][r], o["poses"][-1]) and bits_equal(st["vels"][r], o["vels"][-1]), r
        assert bits_equal(st["force"][r], o["extra"][-1, :, 2:]) and bits_equal(st["dists"][r], o["dists"][-1]), r
        assert np.array_equal(_dense_words(st["coll"][r], E), oracle.coll_to_dense(o["coll"], E)[-1]), r
        assert rows["n_collisions"][r] == o["n_events"], r


# --------------------------------------------------------------------------- wide tiles (E > 64)
def _oracle_one(oracle, packed, r, dt, n_max, **kw):
    from scenario_gym_amd.packing import unpack_scenario

    s = unpack_scenario(packed, r)
    return oracle.rollout(s["knot_off"], s["knots"], s["bbox"], s["etype"], s["kind"], s["ego"], s["t0"], s["length"],
                          dt, ctrl=s["ctrl"], max_steps=n_max, route_off=s.get("route_off"), routes=s.get("routes"), **kw)


def _dense_words(coll, E):
    """[E, W] uint64 rows -> [E, E] adjacency."""
    coll = coll.reshape(E, -1)
    bits = np.unpackbits(np.ascontiguousarray(coll).view(np.uint8), axis=-1, bitorder="little")
    return bits[:, :E]


@pytest.mark.parametrize("R,E,steps,ego_kind", [(24, 100, 150, "pid"), (16, 200, 100, "replay"), (12, 256, 80, "pid"),
                                                (6, 300, 90, "replay"), (5, 512, 70, "pid"), (4, 511, 60, "vehicle")])
def test_wide_scenarios_match_oracle(sga, oracle, R, E, steps, ego_kind):
    """Scenarios of 65..512 entities span 2, 4 or 8 wavefronts of one workgroup: same bits as the oracle."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    kind = dict(replay=L.KIND_AGENT_REPLAY, pid=L.KIND_AGENT_PID, vehicle=L.KIND_AGENT_VEHICLE)[ego_kind]
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=kind, static_frac=0.15, vanish_frac=0.2, extent=50.0 if E <= 256 else 75.0)
    acts = synthetic.make_actions(steps, R, seed=E) if ego_kind == "vehicle" else None
    st, rows, events, t, poses = _engine_run(sga, packed, 1 / 30, steps, ev_cap=256, actions=acts)
    n_ev = 0
    for r in range(R):
        kw = dict(actions=acts[:, r], force_steps=True) if acts is not None else {}
        o = _oracle_one(oracle, packed, r, 1 / 30, steps, **kw)
        n = o["n_steps"]
        assert rows["n_steps"][r] == n and rows["final_t"][r] == o["final_t"], r
        assert bits_equal(poses[: n + 1, r], o["poses"]), r
        assert bits_equal(st["vels"][r], o["vels"][-1]) and bits_equal(st["dists"][r], o["dists"][-1]), r
        assert np.array_equal(_dense_words(st["coll"][r], E), oracle.coll_to_dense(o["coll"], E)[-1]), r
        for k in ("ego_avg_speed", "ego_max_speed", "ego_distance_travelled"):
            assert rows[k][r] == o["metric_" + k], (r, k)
        ev = events[events["scenario"] == r]
        assert np.array_equal(ev["t"], o["ev_t"]) and np.array_equal(ev["other"], o["ev_other"]), r
        n_ev += len(ev)
    assert n_ev > 0


@pytest.mark.parametrize("R,E,steps,side,noise", [(3, 300, 90, 22.0, "off"), (2, 512, 70, 30.0, "device"), (3, 400, 60, 60.0, "off")])
def test_wide_crowds_match_oracle(sga, oracle, R, E, steps, side, noise):
    """Pedestrian agents in scenarios of 257..512 entities: the general pedestrian variant on eight wavefronts of one
    workgroup (rollout_kernel<64, 8, true, false>; the crowd kernels stop at 256).  Poses of every step, forces, distances,
    collision rows (8 words per entity), ego metrics and events equal the oracle's, with the counter-based noise too; a
    vehicle and a replay entity among the pedestrians (every kind in one wide scenario)."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic
    from scenario_gym_amd.packing import unpack_scenario

    packed = synthetic.make_crowd(R, E, n_steps=steps, side=side)
    # entity 1 of every scenario becomes a replay entity (a recorded pedestrian), entity 2 a PID car crossing the square
    kn = packed.knots.reshape(R * E, 2, 7)
    for r in range(R):
        i1, i2 = r * E + 1, r * E + 2
        packed.kind[i1] = L.KIND_REPLAY
        kn[i1, 1, 1:3] = kn[i1, 0, 1:3] + np.array([3.0, -2.0])
        packed.kind[i2] = L.KIND_AGENT_PID
        packed.etype[i2] = 0
        packed.bbox[i2] = synthetic.CAR1_BBOX
        packed.ctrl[i2] = synthetic.DEFAULT_CTRL
        kn[i2, 0, 1:3] = (-side / 2 - 4.0, 0.5 * r)
        kn[i2, 1, 1:3] = (side / 2 + 4.0, 0.5 * r)
        kn[i2, :, 4] = 0.0
    keep = np.ones(R * E, bool)
    keep[[r * E + k for r in range(R) for k in (1, 2)]] = False
    packed.routes = packed.routes.reshape(R * E, 2, 2)[keep].reshape(-1, 2)
    packed.route_off = np.concatenate([[0], np.cumsum(np.where(keep, 2, 0))]).astype(np.int64)
    packed = packed.validate()
    kw, noise_o = {}, [None] * R
    if noise == "device":
        kw = dict(social_force=dict(std_lon=0.1, std_lat=0.05, noise="device", noise_seed=3))
        noise_o = [dict(mode="device", std_lon=0.1, std_lat=0.05, seed=3, scenario_index=r) for r in range(R)]
    eng = sga.RolloutEngine(R, E, record_capacity=steps + 1, event_capacity=512, **kw)
    eng.upload(packed)
    eng.rollout(steps)
    st = eng.state()
    rows, events = eng.metrics()
    t, poses = eng.record(steps + 1)
    eng.close()
    for r in range(R):
        s = unpack_scenario(packed, r)
        o = oracle.rollout(s["knot_off"], s["knots"], s["bbox"], s["etype"], s["kind"], s["ego"], s["t0"], s["length"], 1 / 30,
                           ctrl=s["ctrl"], route_off=s["route_off"], routes=s["routes"], max_steps=steps, event_cap=512, noise=noise_o[r])
        n = o["n_steps"]
        assert rows["n_steps"][r] == n, r
        assert bits_equal(poses[: n + 1, r], o["poses"]), (r, "poses")
        ped = s["kind"] == L.KIND_AGENT_PEDESTRIAN
        assert bits_equal(st["force"][r][ped], o["extra"][-1, ped, 2:]) and bits_equal(st["dists"][r], o["dists"][-1]), r
        assert np.array_equal(_dense_words(st["coll"][r], E), oracle.coll_to_dense(o["coll"], E)[-1]), r
        for k in ("ego_avg_speed", "ego_max_speed", "ego_distance_travelled"):
            assert rows[k][r] == o["metric_" + k], (r, k)
        ev = events[events["scenario"] == r]
        assert np.array_equal(ev["t"], o["ev_t"][: len(ev)]) and np.array_equal(ev["other"], o["ev_other"][: len(ev)]), r


@pytest.mark.parametrize("R,E,steps,ego_kind", [(3, 700, 60, "pid"), (2, 1024, 50, "replay"), (2, 1024, 40, "vehicle"), (1, 1500, 30, "pid")])
def test_scenarios_beyond_512_entities_match_oracle(sga, oracle, R, E, steps, ego_kind):
    """No entity ceiling (sgym_wide.hpp): scenarios of 700, 1024, 1500 entities step as four kernels over as many workgroups
    as they need.  Same checks as the fused widths: poses of every step, velocities, distances, collision rows (E / 64 words
    per entity), ego metrics, events -- replay, PID and external-action egos, static and vanishing entities."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    kind = dict(replay=L.KIND_AGENT_REPLAY, pid=L.KIND_AGENT_PID, vehicle=L.KIND_AGENT_VEHICLE)[ego_kind]
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=kind, static_frac=0.15, vanish_frac=0.2, extent=110.0)
    acts = synthetic.make_actions(steps, R, seed=E) if ego_kind == "vehicle" else None
    st, rows, events, t, poses = _engine_run(sga, packed, 1 / 30, steps, ev_cap=256, actions=acts)
    n_ev = 0
    for r in range(R):
        kw = dict(actions=acts[:, r], force_steps=True) if acts is not None else {}
        o = _oracle_one(oracle, packed, r, 1 / 30, steps, **kw)
        n = o["n_steps"]
        assert rows["n_steps"][r] == n and rows["final_t"][r] == o["final_t"], r
        assert bits_equal(poses[: n + 1, r], o["poses"]), r
        assert bits_equal(st["vels"][r], o["vels"][-1]) and bits_equal(st["dists"][r], o["dists"][-1]), r
        assert np.array_equal(_dense_words(st["coll"][r], E), oracle.coll_to_dense(o["coll"], E)[-1]), r
        for k in ("ego_avg_speed", "ego_max_speed", "ego_distance_travelled"):
            assert rows[k][r] == o["metric_" + k], (r, k)
        ev = events[events["scenario"] == r]
        assert np.array_equal(ev["t"], o["ev_t"]) and np.array_equal(ev["other"], o["ev_other"]), r
        n_ev += len(ev)
    assert n_ev > 0


@pytest.mark.parametrize("persist", [False, True])
def test_caller_run_agents_beyond_512_entities(sga, persist):
    """Custom agents on scenarios of more than 512 entities (sg_set_external_poses + sg_step, one tick at a time): egos whose
    poses the CALLER supplies -- here the poses a ReplayTrajectoryAgent returns, so the whole state equals the same batch with
    device-side replay agents bit for bit --, then `None` (NaN) from some step on: the entity vanishes, or keeps its pose
    under persist (scenario_gym.py:233-239), exactly as at the fused widths."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    R, E, steps, quit_at = 2, 700, 24, 15
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=L.KIND_AGENT_REPLAY, static_frac=0.15, vanish_frac=0.2, extent=110.0)
    ref = sga.RolloutEngine(R, E, event_capacity=256, persist=persist)
    ref.upload(packed)
    packed.kind[0::E] = L.KIND_AGENT_EXTERNAL
    eng = sga.RolloutEngine(R, E, event_capacity=256, persist=persist)
    eng.upload(packed)
    ext = np.full((R, E, 6), np.nan)
    last = None
    for k in range(steps):
        ref.step(1)
        want = ref.state()
        if k < quit_at:
            ext[:, 0] = want["poses"][:, 0]  # what the replay agent returned for this tick
            last = ext[:, 0].copy()
        else:
            ext[:, 0] = np.nan               # the agent returns None
        eng.set_external_poses(ext)
        eng.step(1)
        got = eng.state()
        if k < quit_at:
            for f in ("poses", "vels", "dists"):
                assert bits_equal(got[f], want[f]), (k, f)
            assert np.array_equal(got["present"], want["present"]) and np.array_equal(got["coll"], want["coll"]), k
        else:
            others = np.ones(E, bool)
            others[0] = False
            assert bits_equal(got["poses"][:, others], want["poses"][:, others]), k
            if persist:
                assert got["present"][:, 0].all() and bits_equal(got["poses"][:, 0], last), k
                if k > quit_at:
                    assert (got["vels"][:, 0] == 0).all(), k
            else:
                assert not got["present"][:, 0].any(), k
    assert np.array_equal(eng.metrics()[0]["n_steps"], ref.metrics()[0]["n_steps"])
    eng.close()
    ref.close()


@pytest.mark.parametrize("E,side,noise", [(1024, 45.0, "off"), (600, 30.0, "device")])
def test_crowds_beyond_512_entities_match_oracle(sga, oracle, E, side, noise):
    """... and with every KIND in one scenario of 1024 entities: pedestrian agents (the social force over all pedestrians of
    the scenario, neighbours in entity order), a replay entity, a PID car; forces, collision rows, events equal the oracle's."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic
    from scenario_gym_amd.packing import unpack_scenario

    R, steps = 2, 45
    packed = synthetic.make_crowd(R, E, n_steps=steps, side=side)
    kn = packed.knots.reshape(R * E, 2, 7)
    for r in range(R):
        i1, i2 = r * E + 1, r * E + 2
        packed.kind[i1] = L.KIND_REPLAY
        kn[i1, 1, 1:3] = kn[i1, 0, 1:3] + np.array([3.0, -2.0])
        packed.kind[i2] = L.KIND_AGENT_PID
        packed.etype[i2] = 0
        packed.bbox[i2] = synthetic.CAR1_BBOX
        packed.ctrl[i2] = synthetic.DEFAULT_CTRL
        kn[i2, 0, 1:3] = (-side / 2 - 4.0, 0.5 * r)
        kn[i2, 1, 1:3] = (side / 2 + 4.0, 0.5 * r)
        kn[i2, :, 4] = 0.0
    keep = np.ones(R * E, bool)
    keep[[r * E + k for r in range(R) for k in (1, 2)]] = False
    packed.routes = packed.routes.reshape(R * E, 2, 2)[keep].reshape(-1, 2)
    packed.route_off = np.concatenate([[0], np.cumsum(np.where(keep, 2, 0))]).astype(np.int64)
    packed = packed.validate()
    kw, noise_o = {}, [None] * R
    if noise == "device":
        kw = dict(social_force=dict(std_lon=0.1, std_lat=0.05, noise="device", noise_seed=3))
        noise_o = [dict(mode="device", std_lon=0.1, std_lat=0.05, seed=3, scenario_index=r) for r in range(R)]
    eng = sga.RolloutEngine(R, E, record_capacity=steps + 1, event_capacity=512, **kw)
    eng.upload(packed)
    eng.rollout(steps)
    st = eng.state()
    rows, events = eng.metrics()
    t, poses = eng.record(steps + 1)
    eng.close()
    for r in range(R):
        s = unpack_scenario(packed, r)
        o = oracle.rollout(s["knot_off"], s["knots"], s["bbox"], s["etype"], s["kind"], s["ego"], s["t0"], s["length"], 1 / 30,
                           ctrl=s["ctrl"], route_off=s["route_off"], routes=s["routes"], max_steps=steps, event_cap=512, noise=noise_o[r])
        n = o["n_steps"]
        assert rows["n_steps"][r] == n, r
        assert bits_equal(poses[: n + 1, r], o["poses"]), (r, "poses")
        ped = s["kind"] == L.KIND_AGENT_PEDESTRIAN
        assert bits_equal(st["force"][r][ped], o["extra"][-1, ped, 2:]) and bits_equal(st["dists"][r], o["dists"][-1]), r
        assert np.array_equal(_dense_words(st["coll"][r], E), oracle.coll_to_dense(o["coll"], E)[-1]), r
        for k in ("ego_avg_speed", "ego_max_speed", "ego_distance_travelled"):
            assert rows[k][r] == o["metric_" + k], (r, k)
        ev = events[events["scenario"] == r]
        assert np.array_equal(ev["t"], o["ev_t"][: len(ev)]) and np.array_equal(ev["other"], o["ev_other"][: len(ev)]), r


def test_wide_scenario_collision_terminal(sga, oracle):
    """terminal_conditions=["collision"] across wavefronts (workgroup-wide OR)."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    packed = synthetic.make_batch(8, 130, n_steps=200, ego_kind=L.KIND_AGENT_REPLAY, extent=80.0)
    st, rows, events, t, poses = _engine_run(sga, packed, 1 / 30, 200, terminal=["max_length", "collision"])
    for r in range(8):
        o = _oracle_one(oracle, packed, r, 1 / 30, 200, terminal_mask=3)
        assert rows["n_steps"][r] == o["n_steps"] and bool(rows["done"][r]) == o["is_done"], r
        assert bits_equal(st["poses"][r], o["poses"][-1]), r


# --------------------------------------------------------------------------- pedestrians / social force
def _ped_packed(g, si, max_speed=None):
    from scenario_gym_amd.engine import DEFAULT_CTRL
    from scenario_gym_amd.packing import default_kinds, pack_arrays
    import scenario_gym_amd._lib as L

    sc = scenario_arrays(g, f"loop{si}/scenario")
    E = len(sc["etype"])
    Rt, vdes = g[f"loop{si}/routes"], g[f"loop{si}/vdes"]
    thr = float(g[f"loop{si}/distance_threshold"]) if f"loop{si}/distance_threshold" in g else 1.0  # (agent.py:26: the default)
    kind = default_kinds(E, sc["ego"])
    ctrl = np.tile(DEFAULT_CTRL, (E, 1))
    roff, rows = [0], []
    for i in range(E):
        isped = not np.isnan(vdes[i])
        if isped:
            kind[i] = L.KIND_AGENT_PEDESTRIAN
            ctrl[i, L.C_PED_SPEED_DESIRED], ctrl[i, L.C_PED_RADIUS] = vdes[i], thr
            if max_speed is not None:
                ctrl[i, L.C_PED_MAX_SPEED] = max_speed
            rows.append(Rt[i])
        roff.append(roff[-1] + (len(Rt[i]) if isped else 0))
    sc = dict(sc, route_off=np.array(roff, np.int64), routes=np.concatenate(rows))
    return pack_arrays([sc], kinds=[kind], ctrls=[ctrl]), E


@pytest.mark.parametrize("si", [0, 1])
def test_pedestrian_closed_loops_match_reference(sga, oracle, si):
    """SocialForce closed loops captured from the reference: poses/velocities <= 1e-8 (contract 1e-5),
    goal indices, collisions and CollisionMetric events exact; bit-identical to the oracle."""
    g = load_golden("pedestrian")
    packed, E = _ped_packed(g, si)
    for dtn, dt in (("dt30", 1 / 30), ("dt10", 0.1)):
        p = f"loop{si}/{dtn}"
        n = int(g[p + "/n_steps"])
        st, rows, events, t, poses = _engine_run(sga, packed, dt, n + 2, ev_cap=128)
        assert rows["n_steps"][0] == n and bits_equal(t[: n + 1, 0], g[p + "/t"])
        ref = g[p + "/poses"]
        assert np.array_equal(np.isnan(poses[: n + 1, 0]), np.isnan(ref))
        assert np.nanmax(np.abs(poses[: n + 1, 0] - ref)) < 1e-8
        assert np.nanmax(np.abs(st["vels"][0] - g[p + "/vels"][-1])) < 1e-8
        ex = g[p + "/extra"][-1]
        ped = ~np.isnan(ex[:, 0])
        assert np.array_equal(st["ctrl_state"][0, ped, 1], ex[ped, 1])                    # goal_idx
        assert np.abs(st["ctrl_state"][0, ped, 0] - ex[ped, 0]).max() < 1e-8              # controller speed
        assert np.abs(st["force"][0, ped] - ex[ped, 2:]).max() < 1e-8                     # PedestrianAgent.force
        assert np.array_equal(_dense(st["coll"][0], E), g[p + "/coll"][-1])
        assert np.array_equal(events["t"], g[p + "/ev_t"]) and np.array_equal(events["other"], g[p + "/ev_other"])
        assert (events["type"] == 5).all()
        o = _oracle_one(oracle, packed, 0, dt, n + 2)
        assert bits_equal(poses[: n + 1, 0], o["poses"]) and bits_equal(st["vels"][0], o["vels"][-1])
        assert bits_equal(st["force"][0, ped], o["extra"][-1, ped, 2:])


@pytest.mark.parametrize("si", [0, 1, 2])
def test_pedestrian_noise_closed_loops_match_reference(sga, oracle, si):
    """The random fluctuations of SocialForce._step (social_force.py:106-114) on the device, stream mode: closed loops of
    the reference with std > 0 after np.random.seed(k) (loop 2: the reference's default std, 70 pedestrians = the crowd
    kernel with two wavefronts per scenario).  The device consumes np.random.RandomState(k).standard_normal in the
    reference's order: poses <= 1e-8 of the reference, bit-identical to the oracle, the same number of variates."""
    g = load_golden("ped_noise")
    packed, E = _ped_packed(g, si)
    std_lon, std_lat, seed = g[f"loop{si}/noise"]
    used = int(g[f"loop{si}/variates_used"])
    normals = np.random.RandomState(int(seed)).standard_normal(used + 64)
    p = f"loop{si}/dt30"
    n = int(g[p + "/n_steps"])
    eng = sga.RolloutEngine(1, E, timestep=1 / 30, record_capacity=n + 3, event_capacity=256,
                            social_force=dict(std_lon=std_lon, std_lat=std_lat, noise="stream", normals=normals[None, :]))
    eng.upload(packed)
    eng.rollout(n + 2)
    st = eng.state()
    rows, events = eng.metrics()
    t, poses = eng.record(n + 3)
    assert rows["n_steps"][0] == n and st["noise_pos"][0] == used
    ref = g[p + "/poses"]
    assert np.array_equal(np.isnan(poses[: n + 1, 0]), np.isnan(ref)) and np.nanmax(np.abs(poses[: n + 1, 0] - ref)) < 1e-8
    ex = g[p + "/extra"][-1]
    ped = ~np.isnan(ex[:, 0])
    assert np.array_equal(st["ctrl_state"][0, ped, 1], ex[ped, 1]) and np.abs(st["force"][0, ped] - ex[ped, 2:]).max() < 1e-8
    assert np.array_equal(events["t"], g[p + "/ev_t"]) and np.array_equal(events["other"], g[p + "/ev_other"])
    o = _oracle_one(oracle, packed, 0, 1 / 30, n + 2, noise=dict(mode="stream", std_lon=std_lon, std_lat=std_lat, normals=normals))
    assert bits_equal(poses[: n + 1, 0], o["poses"]) and bits_equal(st["force"][0, ped], o["extra"][-1, ped, 2:])
    # a masked reset rewinds the stream: the same rollout again
    eng.reset()
    eng.rollout(n + 2)
    t2, poses2 = eng.record(n + 3)
    assert bits_equal(poses2, poses)
    # a stream that is too short is an error when the metrics are read, not a silent zero
    eng.set_ped_noise("stream", std_lon, std_lat, normals=normals[None, : used // 2])
    eng.reset()
    eng.rollout(n + 2)
    with pytest.raises(RuntimeError, match="noise variates"):
        eng.metrics()
    eng.close()


def test_headline_horizon_pid_ego_on_the_device(sga, oracle):
    """BASELINE config 3 at its own length against the REAL reference (tests/golden/long.npz: two scenarios of the bench's
    family, PIDAgent ego, 10,000 steps): the device's clock and every replay lane bit for bit over the whole horizon; the PID
    ego inside the 1e-5 contract on every step for as long as the reference agrees with its own one-ulp twin (1,290 / 720
    steps: this closed loop multiplies a rounding error by ten every ~100 steps, test_oracle_golden.check_long_c3), diverging
    no faster than that twin afterwards; and the same bits as the oracle over all 10,000."""
    from test_oracle_golden import check_long_c3, long_c3_batch

    packed, g = long_c3_batch()
    st, rows, events, t, poses = _engine_run(sga, packed, 1 / 30, 10002, ev_cap=64)
    for k in (0, 1):
        n = int(rows["n_steps"][k])
        err, prefix, over = check_long_c3(g, k, n, t[: n + 1, k], poses[: n + 1, k, 0], None, st["poses"][k], st["vels"][k], st["dists"][k])
        o = _oracle_one(oracle, packed, k, 1 / 30, 10002)
        assert bits_equal(poses[: n + 1, k], o["poses"]) and bits_equal(st["ctrl_state"][k, 0], o["extra"][-1, 0])
        print(f"device vs reference c3/{k}: max |ego pose error| on the first {prefix} steps = {err:.3e}, exceeds 1e-5 at step {over}")


def test_headline_horizon_with_a_tracking_pid_on_the_device(sga, oracle):
    """The 1e-5 contract over config 3's full horizon where it is well-posed (long.npz c3t: the bench's scenarios, the PID gains of
    the reference's own controller test, 10,000 steps of the real reference): the device's ego pose after EVERY step, the final
    state of all entities (replay lanes bit for bit), the controller state and the three ego metrics; same bits as the oracle."""
    from test_oracle_golden import check_long_c3t, long_c3_batch

    packed, g = long_c3_batch(tracking=True)
    st, rows, events, t, poses = _engine_run(sga, packed, 1 / 30, 10002, ev_cap=64)
    for k in (0, 1):
        n = int(rows["n_steps"][k])
        err = check_long_c3t(g, k, n, t[: n + 1, k], poses[: n + 1, k, 0], None, st["poses"][k], st["vels"][k], st["dists"][k],
                             {name: float(rows[name][k]) for name in ("ego_avg_speed", "ego_max_speed", "ego_distance_travelled")})
        assert np.abs(st["ctrl_state"][k, 0] - g[f"c3t/{k}/ego"][-1, 6:10]).max() < 1e-8 and err < 1e-8
        o = _oracle_one(oracle, packed, k, 1 / 30, 10002)
        assert bits_equal(poses[: n + 1, k], o["poses"]) and bits_equal(st["ctrl_state"][k, 0], o["extra"][-1, 0])
        print(f"device vs reference c3t/{k}: max |ego pose error| over 10,000 steps = {err:.3e}")


@pytest.mark.parametrize("k", [0, 1])
def test_long_crowd_on_the_device_matches_reference(sga, oracle, k):
    """32 pedestrians, 3,300 steps, recorded from the REAL reference (long.npz; k = 1 with the reference's noise from numpy's
    global stream): every recorded pose within the contract, goal indices, final adjacency and the ego's events exact, the
    same number of variates consumed -- and the same bits as the oracle."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd.engine import DEFAULT_CTRL
    from scenario_gym_amd.packing import pack_arrays
    from test_oracle_golden import check_long_crowd

    g = load_golden("long")
    sc = scenario_arrays(g, f"crowd/{k}/scenario")
    E = len(sc["etype"])
    ctrl = np.tile(DEFAULT_CTRL, (E, 1))
    ctrl[:, L.C_PED_SPEED_DESIRED], ctrl[:, L.C_PED_RADIUS] = g[f"crowd/{k}/vdes"], float(g[f"crowd/{k}/distance_threshold"])
    nw = g[f"crowd/{k}/routes"].shape[1]
    sc = dict(sc, route_off=np.arange(E + 1, dtype=np.int64) * nw, routes=g[f"crowd/{k}/routes"].reshape(-1, 2))
    packed = pack_arrays([sc], kinds=[np.full(E, L.KIND_AGENT_PEDESTRIAN, np.int32)], ctrls=[ctrl])
    std_lon, std_lat, seed = g[f"crowd/{k}/noise"]
    kw, noise = {}, None
    if std_lon or std_lat:
        used = int(g[f"crowd/{k}/variates_used"])
        normals = np.random.RandomState(int(seed)).standard_normal(used + 64)
        kw = dict(social_force=dict(std_lon=std_lon, std_lat=std_lat, noise="stream", normals=normals[None, :]))
        noise = dict(mode="stream", std_lon=std_lon, std_lat=std_lat, normals=normals)
    n_cap = 3304
    eng = sga.RolloutEngine(1, E, timestep=1 / 30, record_capacity=n_cap, event_capacity=512, **kw)
    eng.upload(packed)
    eng.rollout(n_cap - 1)
    st = eng.state()
    rows, events = eng.metrics()
    t, poses = eng.record(n_cap)
    eng.close()
    n = int(rows["n_steps"][0])
    if noise:
        assert st["noise_pos"][0] == used
    extra = np.concatenate([st["ctrl_state"][0][:, :2], st["force"][0]], axis=1)
    err = check_long_crowd(g, k, n, t[: n + 1, 0], poses[: n + 1, 0], st["vels"][0], st["dists"][0], extra, _dense_words(st["coll"][0], E),
                           events["t"], events["other"])
    o = _oracle_one(oracle, packed, 0, 1 / 30, n_cap - 1, noise=noise, event_cap=512)
    assert bits_equal(poses[: n + 1, 0], o["poses"]) and bits_equal(st["force"][0], o["extra"][-1, :, 2:])
    print(f"device vs reference, crowd/{k} over 3,300 steps: max |pose error| = {err:.3e}")


@pytest.mark.parametrize("si", [0, 1, 2, 3])
def test_random_walk_closed_loops_match_reference(sga, oracle, si):
    """RandomWalk (pedestrian/random_walk.py:22-44) on the device, stream mode: closed loops of the reference's
    PedestrianAgent(..., behaviour=RandomWalk(params)) after np.random.seed(k) -- with a bias and a clipping max_speed (loop
    1), the reference's default parameters (loop 2), std 0 (loop 3).  Poses <= 1e-8 of the reference, goal indices and
    controller speeds, the (zero) force, events; bit-identical to the oracle; the same number of variates."""
    g = load_golden("random_walk")
    std_lon, std_lat, bias_lon, bias_lat, max_speed, seed = g[f"loop{si}/params"]
    packed, E = _ped_packed(g, si, max_speed=max_speed)
    used = int(g[f"loop{si}/variates_used"])
    normals = np.random.RandomState(int(seed)).standard_normal(used + 64)
    p = f"loop{si}/dt30"
    n = int(g[p + "/n_steps"])
    eng = sga.RolloutEngine(1, E, timestep=1 / 30, record_capacity=n + 3, event_capacity=256,
                            social_force=dict(behaviour="random_walk", bias_lon=bias_lon, bias_lat=bias_lat, std_lon=std_lon,
                                              std_lat=std_lat, noise="stream", normals=normals[None, :]))
    eng.upload(packed)
    with pytest.raises(RuntimeError, match="before sg_upload"):
        eng.set_ped_behaviour("social_force")
    eng.rollout(n + 2)
    st = eng.state()
    rows, events = eng.metrics()
    t, poses = eng.record(n + 3)
    eng.close()
    assert rows["n_steps"][0] == n and st["noise_pos"][0] == used
    ref = g[p + "/poses"]
    assert np.array_equal(np.isnan(poses[: n + 1, 0]), np.isnan(ref)) and np.nanmax(np.abs(poses[: n + 1, 0] - ref)) < 1e-8
    ex = g[p + "/extra"][-1]
    ped = ~np.isnan(ex[:, 0])
    assert np.array_equal(st["ctrl_state"][0, ped, 1], ex[ped, 1]) and np.abs(st["ctrl_state"][0, ped, 0] - ex[ped, 0]).max() < 1e-8
    assert not st["force"][0, ped].any()
    assert np.array_equal(events["t"], g[p + "/ev_t"]) and np.array_equal(events["other"], g[p + "/ev_other"])
    sf = oracle.social_force_params(bias_lon=bias_lon, bias_lat=bias_lat)
    o = _oracle_one(oracle, packed, 0, 1 / 30, n + 2, sf=sf, behaviour="random_walk",
                    noise=dict(mode="stream", std_lon=std_lon, std_lat=std_lat, normals=normals))
    assert bits_equal(poses[: n + 1, 0], o["poses"]) and bits_equal(st["vels"][0], o["vels"][-1])
    assert bits_equal(st["ctrl_state"][0, ped, 0], o["extra"][-1, ped, 0])


@pytest.mark.parametrize("R,E,steps,side,noise", [(8, 256, 60, 30.0, "device"), (16, 40, 90, 10.0, "off"), (3, 300, 50, 30.0, "device"),
                                                  (2, 700, 40, 40.0, "device")])
def test_random_walk_crowds_match_oracle(sga, oracle, R, E, steps, side, noise):
    """RandomWalk over the crowd batches of config 5's family -- 40 ... 700 pedestrians per scenario: the general pedestrian
    variant on one to eight wavefronts and the multi-kernel step beyond 512 entities (the crowd kernels hold the social force
    model only) -- with the counter-based generator and without noise: every recorded pose, the final state, the events
    equal the oracle's; the walk differs from the social force model's."""
    from scenario_gym_amd import synthetic

    packed = synthetic.make_crowd(R, E, n_steps=steps, side=side)
    kw = dict(behaviour="random_walk", bias_lon=0.05, bias_lat=-0.02)
    noise_of = None
    if noise == "device":
        kw.update(std_lon=0.2, std_lat=0.3, noise="device", noise_seed=77)
        noise_of = lambda r: dict(mode="device", std_lon=0.2, std_lat=0.3, seed=77, scenario_index=r)  # noqa: E731
    eng = sga.RolloutEngine(R, E, record_capacity=steps + 1, event_capacity=512, social_force=kw)
    eng.upload(packed)
    eng.rollout(steps)
    st = eng.state()
    t, poses = eng.record(steps + 1)
    from oracle import check

    sf = oracle.social_force_params(bias_lon=0.05, bias_lat=-0.02)
    ver = check.verify_engine(eng, packed, 1 / 30, steps, K=R, event_cap=512, ped=True, sf=sf, noise_of=noise_of, behaviour="random_walk")
    assert ver["equal"], ver["mismatches"]
    eng.close()
    o = _oracle_one(oracle, packed, 0, 1 / 30, steps, sf=sf, behaviour="random_walk", noise=None if noise_of is None else noise_of(0))
    assert bits_equal(poses[: o["n_steps"] + 1, 0], o["poses"])
    assert not st["force"].any()
    o0 = _oracle_one(oracle, packed, 0, 1 / 30, steps, sf=sf, noise=None if noise_of is None else noise_of(0))
    assert np.nanmax(np.abs(o0["poses"][-1] - o["poses"][-1])) > 1e-3


@pytest.mark.parametrize("R,E,steps,side", [(6, 256, 70, 30.0), (16, 40, 100, 10.0), (5, 100, 60, 14.0)])
def test_crowd_with_device_noise_matches_oracle(sga, oracle, R, E, steps, side):
    """Noise mode "device" (timing / production runs): the counter-based generator -- Philox4x32-10 keyed by (seed,
    scenario) at counter (entity, step), Box-Muller with the shared log / sin / cos -- is restated in the oracle: crowds
    walk the same bits, and differently from the noise-free run."""
    from scenario_gym_amd import synthetic

    packed = synthetic.make_crowd(R, E, n_steps=steps, side=side)
    kw = dict(std_lon=0.2, std_lat=0.1, noise="device", noise_seed=0xC0FFEE1234)
    eng = sga.RolloutEngine(R, E, record_capacity=steps + 1, event_capacity=512, social_force=kw)
    eng.upload(packed)
    eng.rollout(steps)
    st = eng.state()
    t, poses = eng.record(steps + 1)
    eng.close()
    for r in range(R):
        o = _oracle_one(oracle, packed, r, 1 / 30, steps,
                        noise=dict(mode="device", std_lon=0.2, std_lat=0.1, seed=0xC0FFEE1234, scenario_index=r))
        assert bits_equal(poses[: o["n_steps"] + 1, r], o["poses"]), r
        assert bits_equal(st["force"][r], o["extra"][-1, :, 2:]), r
    o0 = _oracle_one(oracle, packed, 0, 1 / 30, steps)
    assert np.abs(o0["poses"][-1] - poses[steps, 0]).max() > 1e-3


@pytest.mark.parametrize("R,E,steps,side", [(32, 40, 120, 12.0), (6, 256, 60, 30.0), (10, 100, 80, 20.0)])
def test_crowd_matches_oracle(sga, oracle, R, E, steps, side):
    """BASELINE config 5 family (all-pedestrian crowds, up to 256 per scenario): bit-identical to the oracle."""
    from scenario_gym_amd import synthetic

    packed = synthetic.make_crowd(R, E, n_steps=steps, side=side)
    st, rows, events, t, poses = _engine_run(sga, packed, 1 / 30, steps, ev_cap=512)
    for r in range(R):
        o = _oracle_one(oracle, packed, r, 1 / 30, steps)
        n = o["n_steps"]
        assert rows["n_steps"][r] == n, r
        assert bits_equal(poses[: n + 1, r], o["poses"]), r
        assert bits_equal(st["vels"][r], o["vels"][-1]) and bits_equal(st["dists"][r], o["dists"][-1]), r
        assert bits_equal(st["force"][r], o["extra"][-1, :, 2:]) and np.array_equal(st["ctrl_state"][r, :, 1], o["extra"][-1, :, 1])
        assert np.array_equal(_dense_words(st["coll"][r], E), oracle.coll_to_dense(o["coll"], E)[-1]), r
        ev = events[events["scenario"] == r]
        assert rows["n_collisions"][r] == o["n_events"]
        assert np.array_equal(ev["t"], o["ev_t"]) and np.array_equal(ev["other"], o["ev_other"]), r
    assert (rows["n_collisions"] > 0).any()


@pytest.mark.parametrize("sf", [
    dict(ped_attract_C=0.05),                                   # attraction on: both sight weights are evaluated
    dict(sight_weight_use=False, ped_attract_C=0.02),           # reference adds attraction before repulsion
    dict(relaxation_time=0.8, ped_repulse_V=2.0, ped_repulse_sigma=0.7, sight_weight=0.3, sight_angle=120),
])
def test_crowd_with_non_default_social_force(sga, oracle, sf):
    """The generic neighbour terms (head rotation != 0, attraction, sight weights on/off, other constants): the
    default-parameter shortcuts of the kernel must not be taken, results stay bit-identical to the oracle."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    R, E, steps = 12, 48, 90
    packed = synthetic.make_crowd(R, E, n_steps=steps, side=10.0)
    rng = np.random.default_rng(11)
    packed.ctrl[:, L.C_PED_HEAD_ROT] = rng.uniform(-0.6, 0.6, R * E)  # PedestrianAgent(head_rot_angle=...)
    eng = sga.RolloutEngine(R, E, record_capacity=steps + 1, event_capacity=256, social_force=sf)
    eng.upload(packed)
    eng.rollout(steps)
    st = eng.state()
    rows, events = eng.metrics()
    t, poses = eng.record(steps + 1)
    eng.close()
    for r in range(R):
        o = _oracle_one(oracle, packed, r, 1 / 30, steps, sf=oracle.social_force_params(**sf))
        n = o["n_steps"]
        assert rows["n_steps"][r] == n, r
        assert bits_equal(poses[: n + 1, r], o["poses"]), r
        assert bits_equal(st["force"][r], o["extra"][-1, :, 2:]), r
        assert np.array_equal(_dense_words(st["coll"][r], E), oracle.coll_to_dense(o["coll"], E)[-1]), r


def test_pedestrian_stepwise_equals_single_launch(sga):
    """gym.step() one launch per step reproduces the single-launch rollout (neighbour candidates and
    goal indices survive the kernel boundary)."""
    from scenario_gym_amd import synthetic

    packed = synthetic.make_crowd(8, 24, n_steps=80, side=8.0)  # nobody reaches max_length in 40 steps
    a = sga.RolloutEngine(8, 24)
    a.upload(packed)
    a.rollout(40)
    b = sga.RolloutEngine(8, 24)
    b.upload(packed)
    for _ in range(40):
        b.step(1)
    sa, sb = a.state(), b.state()
    assert (sa["n_steps"] == 40).all() and (sb["n_steps"] == 40).all()
    for k in ("poses", "vels", "dists", "force", "ctrl_state"):
        assert bits_equal(sa[k], sb[k]), k
    assert np.array_equal(sa["coll"], sb["coll"])
    a.close()
    b.close()


# --------------------------------------------------------------------------- edge cases
def _mini(knots_per_entity, bbox=(2.0, 4.2, 1.37, 0.0), etype=2, length=None, t0=None, ego=0):
    off = np.concatenate([[0], np.cumsum([len(k) for k in knots_per_entity])]).astype(np.int64)
    knots = np.concatenate([np.asarray(k, np.float64) for k in knots_per_entity])
    E = len(knots_per_entity)
    return dict(knot_off=off, knots=knots, bbox=np.tile(bbox, (E, 1)), etype=np.full(E, etype, np.int32), ego=ego,
                t0=max(0.0, float(knots[off[ego], 0])) if t0 is None else t0,
                length=float(max(k[-1][0] for k in knots_per_entity)) if length is None else length)


def _row(t, x, y, h=0.0):
    return [t, x, y, 0.0, h, 0.0, 0.0]


def _check_against_oracle(sga, oracle, scs, dt, n_max, **kw):
    from scenario_gym_amd.packing import default_kinds, pack_arrays

    packed = pack_arrays(scs)
    st, rows, events, t, poses = _engine_run(sga, packed, dt, n_max, **kw)
    for r, sc in enumerate(scs):
        E = len(sc["etype"])
        o = oracle.rollout(sc["knot_off"], sc["knots"], sc["bbox"], sc["etype"], default_kinds(E, sc["ego"]), sc["ego"],
                           sc["t0"], sc["length"], dt, max_steps=n_max, persist=kw.get("persist", False))
        n = o["n_steps"]
        assert rows["n_steps"][r] == n and bool(rows["done"][r]) == o["is_done"], r
        assert bits_equal(poses[: n + 1, r, :E], o["poses"]), r
        assert bits_equal(st["vels"][r, :E], o["vels"][-1]) and bits_equal(st["dists"][r, :E], o["dists"][-1]), r
        assert np.array_equal(st["coll"][r, :E], o["coll"][-1, :, 0]), r
        assert np.isnan(poses[: n + 1, r, E:]).all()  # padding slots are never present
    return st, rows, events


def test_degenerate_and_ragged_scenarios(sga, oracle):
    """One-entity scenario, all-static scenario, zero-length scenario, an entity that ended before t0,
    an ego that starts late, ragged widths padded in one batch (tile of 8 lanes, partial last wavefront)."""
    scs = [
        _mini([[_row(0, 0, 0), _row(3, 30, 0)]]),                                             # ego alone
        _mini([[_row(0, 0, 0)], [_row(0, 1, 0)], [_row(2, 50, 50)]], length=1.0),             # everything static
        _mini([[_row(1.0, 5, 5)]], length=1.0),                                               # zero length: one step
        _mini([[_row(2, 0, 0), _row(4, 10, 0)], [_row(0, 3, 0), _row(1, 4, 0)], [_row(0, 9, 9), _row(9, 0, 0)]]),  # #1 ended before t0 = 2
        _mini([[_row(0, 0, 0), _row(2, 5, 1)], [_row(0.5, 2, 0.5), _row(1.5, 2, 0.4)], [_row(0, 2.5, 0.6)],
               [_row(0, -3, 0), _row(2, 8, 1.2)], [_row(1.0, 4, 1), _row(1.9, 4, 1.1)]]),    # 5 wide: crossings, vanishing
    ]
    for persist in (False, True):
        _check_against_oracle(sga, oracle, scs, 0.1, 100, persist=persist)
    _check_against_oracle(sga, oracle, scs * 3, 1 / 30, 300)  # 15 scenarios x 8 lanes = partial second wavefront


def test_far_from_origin_uses_the_all_pairs_fallback(sga, oracle):
    """Coordinates beyond 4000 broad-phase cells (~4e4 m) leave the stripe masks for the all-pairs path;
    beyond 1e5 rad the shared sin/cos defers to libm/ocml (not bit-pinned): stay below that."""
    rng = np.random.default_rng(5)
    scs = []
    for k, origin in enumerate((3.0e4, 2.0e5, -7.5e5)):
        ents = []
        for i in range(12):
            x0, y0 = origin + rng.uniform(-15, 15), -origin + rng.uniform(-15, 15)
            ents.append([_row(0, x0, y0, 0.3 * i), _row(4, x0 + rng.uniform(-20, 20), y0 + rng.uniform(-20, 20), 0.3 * i + 1)])
        scs.append(_mini(ents))
    st, rows, events = _check_against_oracle(sga, oracle, scs, 1 / 30, 130, ev_cap=256)
    assert (st["coll"] != 0).any()


def test_sliced_path_orders_its_two_streams(sga, monkeypatch):
    """The time-sliced path finishes on two streams: the launch that materialises the last executed step (it starts from the
    scenario records of the reset) and the per-scenario ordered pass (it overwrites those records with the final ones).
    SG_SLICE_DELAY_US holds the first one back by a millisecond: the second must still wait for it.  (It did not until round
    5 -- a scenario already marked done sat the last launch out and kept its reset poses, on one suite run in four of a
    cold box.)"""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    out = []
    for delay in ("0", "1000"):
        monkeypatch.setenv("SG_SLICE_DELAY_US", delay)
        for ego, E, term in ((L.KIND_AGENT_REPLAY, 31, ["max_length", "ego_collision"]), (L.KIND_AGENT_VEHICLE, 16, ["max_length", "collision"])):
            packed = synthetic.make_batch(12, E, n_steps=90, ego_kind=ego, static_frac=0.15, vanish_frac=0.25, extent=14.0)
            res = []
            for slicing in (False, "always"):
                eng = sga.RolloutEngine(12, E, terminal_conditions=term, event_capacity=8)
                eng.set_slicing(slicing)
                eng.upload(packed)
                res.append(_final_results(eng, 90))
                eng.close()
            (sa, ra, ea), (sb, rb, eb) = res
            for k in ("poses", "vels", "dists", "t", "prev_t", "ctrl_state"):
                assert bits_equal(sa[k], sb[k]), (delay, E, k)
            assert np.array_equal(sa["coll"], sb["coll"]) and np.array_equal(ra, rb) and np.array_equal(ea, eb), (delay, E)
            out.append(int(ra["n_steps"].min()))
    assert min(out) < 90  # (some scenario ended early: its last step is not the call's last)


@pytest.mark.parametrize("byte", [255, 127])
def test_poisoned_allocations_change_nothing(byte):
    """SG_POISON fills every device array the library hands out WITHOUT zeroing it (NaNs / -1 with 255, huge integers with
    127) instead of leaving what the allocator found: a read of something no kernel wrote shows in a fresh process as it would
    after a thousand other handles.  A cross-section of the parity tests (table path, time-sliced path, crowds, RSS, the
    multi-kernel step) runs in a child process under it."""
    import subprocess

    env = dict(os.environ, SG_POISON=str(byte))
    k = ("test_synthetic_batch_matches_oracle or test_sliced_rollout_with_controlled_lanes or test_crowd_matches_oracle or "
         "test_rss_fused_rollout_matches_oracle or test_scenarios_beyond_512_entities_match_oracle or test_queue_launch_equals_chunk_launches")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", "-k", k],
                         capture_output=True, text=True, timeout=900, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stdout[-3000:]
    assert " passed" in out.stdout and "failed" not in out.stdout


def test_abi_rejects_bad_input(sga):
    import scenario_gym_amd._lib as L
    from scenario_gym_amd.packing import pack_arrays

    with pytest.raises(RuntimeError, match="n_entities"):
        sga.RolloutEngine(4, 16385)
    eng = sga.RolloutEngine(1, 8)
    with pytest.raises(RuntimeError, match="model_of"):   # (a refused call leaves the handle as it was)
        eng.set_ped_models([dict(behaviour="social_force"), dict(behaviour="random_walk")], np.full(8, 2, np.int32))
    eng.close()
    with pytest.raises(RuntimeError, match="timestep"):
        sga.RolloutEngine(4, 4, timestep=0.0)
    with pytest.raises(ValueError):
        sga.RolloutEngine(4, 4, terminal_conditions=["ego_in_the_air"])
    good = _mini([[_row(0, 0, 0), _row(1, 1, 0)], [_row(0, 5, 0)]])
    eng = sga.RolloutEngine(1, 2)
    with pytest.raises(RuntimeError, match="no scenarios uploaded"):
        eng.rollout(3)
    tri = dict(ring_off=[0, 1], vert_off=[0, 3], verts=[[0, 0], [1, 0], [0, 1]], layers=[1])
    with pytest.raises(RuntimeError, match="no scenarios uploaded"):
        eng.set_road_networks([tri], [0])
    eng.upload(pack_arrays([good]))
    with pytest.raises(RuntimeError, match="out of range"):
        eng.set_road_networks([tri], [1])
    with pytest.raises(RuntimeError, match="not finite"):
        eng.set_road_networks([dict(tri, verts=[[0, 0], [np.nan, 0], [0, 1]])], [0])
    with pytest.raises(RuntimeError, match="SG_LAYER"):
        eng.raster_map([3])
    eng.set_road_networks([], [-1])  # no networks at all: empty surfaces
    assert not eng.raster_map([1, 2]).any()
    eng.close()
    eng = sga.RolloutEngine(1, 2)
    bad = pack_arrays([good])
    bad.kind = bad.kind.copy()
    bad.kind[1] = 9
    with pytest.raises(RuntimeError, match="unknown"):
        eng.upload(bad)
    bad = pack_arrays([good])
    bad.knots = bad.knots.copy()
    bad.knots[1, 0] = 0.0  # not strictly increasing
    with pytest.raises(RuntimeError, match="strictly increasing"):
        eng.upload(bad)
    bad = pack_arrays([good])
    bad.ego = np.array([5], np.int32)
    with pytest.raises(RuntimeError, match="ego"):
        eng.upload(bad)
    bad = pack_arrays([good])
    bad.kind = bad.kind.copy()
    bad.kind[0] = L.KIND_AGENT_PEDESTRIAN  # pedestrian agent without a route
    with pytest.raises(RuntimeError, match="route"):
        eng.upload(bad)
    eng.upload(pack_arrays([good]))  # the handle is still usable
    eng.rollout(50)
    assert eng.state()["done"].all()
    with pytest.raises(RuntimeError, match="record_capacity"):
        eng.record(5)
    eng.close()


# --------------------------------------------------------------------------- launch policy / fp32 trig
INLINE = dict(tab_min_steps=1 << 30)                       # controllers inside the rollout kernel
TABLE_SMALL_CHUNKS = dict(tab_min_steps=1, chunk_steps=7)  # controller pre-pass, many chunk boundaries, 2 streams
TABLE_SERIAL = dict(tab_min_steps=1, chunk_steps=64, overlap=0)


@pytest.mark.parametrize("ego_kind,terminal", [("pid", None), ("vehicle", None), ("pid", ["max_length", "ego_collision"])])
def test_controller_prepass_equals_inline_controllers(sga, ego_kind, terminal):
    """control_kernel + rollout_kernel<TAB> (any chunking, with or without stream overlap) produce the bits of the
    single-kernel path: poses of every step, final state, controller state, metrics, events."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    R, E, steps = 96, 24, 150
    kind = dict(pid=L.KIND_AGENT_PID, vehicle=L.KIND_AGENT_VEHICLE)[ego_kind]
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=kind, static_frac=0.15, vanish_frac=0.2, extent=25.0)
    acts = synthetic.make_actions(steps, R) if ego_kind == "vehicle" else None
    runs = [_engine_run(sga, packed, 1 / 30, steps, terminal=terminal, actions=acts, ev_cap=128, tuning=tn)
            for tn in (INLINE, TABLE_SMALL_CHUNKS, TABLE_SERIAL)]
    st0, rows0, ev0, t0, poses0 = runs[0]
    if terminal:
        assert (rows0["n_steps"] < steps).any() and (rows0["n_steps"] == steps).any()  # some stop early, some do not
    for st, rows, ev, t, poses in runs[1:]:
        for k in ("poses", "vels", "dists", "ctrl_state", "t", "prev_t"):
            assert bits_equal(st[k], st0[k]), k
        assert np.array_equal(st["coll"], st0["coll"]) and np.array_equal(st["present"], st0["present"])
        assert rows.tobytes() == rows0.tobytes() and ev.tobytes() == ev0.tobytes()
        assert bits_equal(t, t0) and bits_equal(poses, poses0)


SCHED_CHUNKS, SCHED_QUEUE = 1, 2  # sg_schedule_info()[0]: chunk launches / one persistent launch (sgym_queue.hpp)


@pytest.mark.parametrize("E,ego_kind,zpr", [(64, "pid", False), (24, "vehicle", False), (12, "pid", False), (64, "pid", True), (30, "pid", True)])
def test_queue_launch_equals_chunk_launches(sga, monkeypatch, E, ego_kind, zpr):
    """The table path runs as ONE persistent launch (rollout_kernel_tabq: pre-pass and rollout roles in one grid, work items
    (chunk, block) from a device-side counter, the table ring as large as the call) -- and, with SG_QUEUE=0, as the chunk
    launches of rounds 1-4 on two streams.  Same bits either way (every recorded pose, final state, controller state,
    metrics, events), with many chunk boundaries, with a table ring of two buffers that the pre-pass has to wait for, for the
    planar and the general table kernel, 64-lane and narrow tiles, rollouts and forced steps with external actions, and a
    rollout continued in pieces.  Which schedule ran is what the handle says it ran: nothing here depends on timing."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    R, steps = 203, 150
    kind = dict(pid=L.KIND_AGENT_PID, vehicle=L.KIND_AGENT_VEHICLE)[ego_kind]
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=kind, static_frac=0.15, vanish_frac=0.2, extent=25.0 if E > 8 else 8.0)
    if zpr:  # knots with z / pitch / roll: the general table kernel (rollout_kernel_tab<G>)
        rng = np.random.default_rng(5)
        packed.knots[:, 3] = rng.normal(0.0, 1.0, len(packed.knots))
        packed.knots[:, 5] = rng.normal(0.0, 0.1, len(packed.knots))
    acts = synthetic.make_actions(steps, R) if ego_kind == "vehicle" else None
    runs, pieces, sched = [], [], []
    for queue, ring in (("0", "0"), ("1", "0"), ("1", "2")):
        monkeypatch.setenv("SG_QUEUE", queue)
        monkeypatch.setenv("SG_QUEUE_RING", ring)
        runs.append(_engine_run(sga, packed, 1 / 30, steps, terminal=["max_length", "ego_collision"], actions=acts, ev_cap=128,
                                tuning=dict(tab_min_steps=1, chunk_steps=16)))
        # ... and continued in pieces
        eng = sga.RolloutEngine(R, E, terminal_conditions=["max_length"], event_capacity=64)
        eng.set_tuning(tab_min_steps=1, chunk_steps=16)
        eng.upload(packed)
        eng.rollout(40)
        eng.rollout_async(70, do_reset=False)
        eng.step(30)
        pieces.append((eng.state(), eng.metrics()))
        sched.append(eng.schedule_info())
        eng.close()
    assert [s["schedule"] for s in sched] == [SCHED_CHUNKS, SCHED_QUEUE, SCHED_QUEUE], sched
    assert sched[1]["ring"] == sched[1]["chunks"] >= 2 and sched[2]["ring"] == 2 and sched[2]["launches"] == 1, sched
    st0, rows0, ev0, t0, poses0 = runs[0]
    if acts is None:
        assert (rows0["n_steps"] < steps).any() and (rows0["n_steps"] == steps).any()
    for st, rows, ev, t, poses in runs[1:]:
        for k in ("poses", "vels", "dists", "ctrl_state", "t", "prev_t"):
            assert bits_equal(st[k], st0[k]), k
        assert np.array_equal(st["coll"], st0["coll"]) and np.array_equal(st["present"], st0["present"])
        assert rows.tobytes() == rows0.tobytes() and ev.tobytes() == ev0.tobytes()
        assert bits_equal(t, t0) and bits_equal(poses, poses0)
    for st, (rows, ev) in pieces[1:]:
        for k in ("poses", "vels", "dists", "ctrl_state", "t", "prev_t"):
            assert bits_equal(st[k], pieces[0][0][k]), k
        assert rows.tobytes() == pieces[0][1][0].tobytes() and ev.tobytes() == pieces[0][1][1].tobytes()


@pytest.mark.parametrize("rss", [False, True])
def test_queue_handoff_variants_agree(sga, monkeypatch, rss):
    """A block's consecutive chunks run on different XCDs, whose L2s are not coherent.  The default hand-over re-stores the
    block's state rows, its scenarios' records and (RSS) the RSS words with write-through stores (SG_QUEUE_HANDOFF=1); the
    other one is a release fence per item (SG_QUEUE_HANDOFF=0).  Same batch through both, with the table ring as large as the
    call and with two buffers, and through the chunk launches: every state row, controller state, metric row, event and
    (RSS) every RSS record are the same bits -- whatever an item writes that a later item of its block reads travels."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    R, E, steps = 700, 64, 260
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=L.KIND_AGENT_PID, static_frac=0.1, vanish_frac=0.3, extent=22.0)
    out, sched = [], []
    for queue, handoff, ring in (("0", "1", "0"), ("1", "1", "0"), ("1", "0", "0"), ("1", "1", "2"), ("1", "0", "2")):
        monkeypatch.setenv("SG_QUEUE", queue)
        monkeypatch.setenv("SG_QUEUE_HANDOFF", handoff)
        monkeypatch.setenv("SG_QUEUE_RING", ring)
        eng = sga.RolloutEngine(R, E, terminal_conditions=["max_length"], event_capacity=96)
        eng.set_tuning(tab_min_steps=1, chunk_steps=16)
        if rss:
            eng.set_rss(True)
        eng.upload(packed)
        eng.rollout(steps)
        m, ev = eng.metrics()
        out.append((eng.state(), m, ev, eng.rss() if rss else None))
        sched.append(eng.schedule_info()["schedule"])
        eng.close()
    assert sched == [SCHED_CHUNKS] + [SCHED_QUEUE] * 4, sched
    st0, m0, ev0, rss0 = out[0]
    assert len(ev0) > 0
    for st, m, ev, rs in out[1:]:
        for k in st0:
            assert np.array_equal(st[k], st0[k], equal_nan=True), k
        assert m.tobytes() == m0.tobytes() and ev.tobytes() == ev0.tobytes()
        if rss:
            for x, y in zip(rs, rss0):
                assert np.array_equal(x, y, equal_nan=True)


def test_queue_give_up_is_loud_and_sticky(sga, monkeypatch):
    """A persistent launch whose wavefronts wait longer than the limit for each other gives up instead of hanging (here on
    demand: a limit of one microsecond, which the first wait for the controller pre-pass exceeds).  The state of the batch is
    undefined from then on, and every call that would run or read it says so -- the synchronising call, a read of the metrics, a
    second rollout queued behind it without a look in between, a continued rollout -- until sg_reset (or a rollout that
    resets, or sg_upload) starts the batch anew; then the same handle runs the batch to the oracle's bits, as chunk launches."""
    import scenario_gym_amd._lib as L
    from oracle import check
    from scenario_gym_amd import synthetic

    R, E, steps, dt = 1024, 64, 400, 1 / 30
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=L.KIND_AGENT_PID, extent=30.0)
    eng = sga.RolloutEngine(R, E, timestep=dt, terminal_conditions=["max_length"], event_capacity=64)
    eng.upload(packed)
    monkeypatch.setenv("SG_QUEUE_TIMEOUT_US", "1")
    eng.rollout_async(steps, do_reset=True)
    eng.rollout_async(steps, do_reset=False)  # (queued behind it: its own slot for the give-up code)
    with pytest.raises(RuntimeError, match="gave up"):
        eng.synchronize()
    assert eng.schedule_info()["schedule"] == SCHED_QUEUE
    with pytest.raises(RuntimeError, match="gave up"):
        eng.metrics()
    with pytest.raises(RuntimeError, match="gave up"):
        eng.rollout_async(10, do_reset=False)
    with pytest.raises(RuntimeError, match="gave up"):
        eng.terminal_flags()
    monkeypatch.delenv("SG_QUEUE_TIMEOUT_US")
    eng.reset()
    eng.rollout_async(steps, do_reset=False)
    eng.synchronize()
    ver = check.verify_engine(eng, packed, dt, steps, K=8, event_cap=64)
    assert ver["equal"], ver["mismatches"]
    # whatever kept the wavefronts from meeting would do so again: after a give-up the handle takes the chunk launches
    assert eng.schedule_info()["schedule"] == SCHED_CHUNKS
    eng.close()


def test_queue_launch_with_many_blocks_per_slot(sga, oracle, monkeypatch):
    """More blocks than the device has wavefront slots (3 x 1024 SIMDs), a ring of three buffers, chunks of 32 steps: every slot
    serves several blocks, the blocks of a chunk finish out of order, the pre-pass waits for the ring -- the state after the
    rollout equals the oracle's on scenarios spread over the batch."""
    import scenario_gym_amd._lib as L
    from oracle import check
    from scenario_gym_amd import synthetic

    R, E, steps, dt = 3600, 64, 200, 1 / 30
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=L.KIND_AGENT_PID, static_frac=0.1, vanish_frac=0.3, extent=30.0)
    monkeypatch.setenv("SG_QUEUE_RING", "3")
    eng = sga.RolloutEngine(R, E, timestep=dt, terminal_conditions=["max_length"], event_capacity=64)
    eng.set_tuning(tab_min_steps=1, chunk_steps=32)
    eng.upload(packed)
    eng.rollout(steps)
    info = eng.schedule_info()
    assert info["schedule"] == SCHED_QUEUE and info["ring"] == 3 and info["chunks"] >= 6 and info["launches"] == 1, info
    ver = check.verify_engine(eng, packed, dt, steps, K=24, event_cap=64)
    assert ver["equal"], ver["mismatches"]
    eng.close()


@pytest.mark.parametrize("queue", ["0", "1"])
@pytest.mark.parametrize("persist", [False, True])
def test_planar_table_kernel_with_late_spawns(sga, oracle, monkeypatch, queue, persist):
    """The planar table kernel (rollout_kernel_tab_planar) never stores the z / pitch / roll rows of a pose: it relies on the
    reset having left +0.0 there for every lane that is absent or spawns later.  A planar batch in which half of the
    entities appear late or vanish (PID ego), as chunk launches and as the persistent launch, with and without persist, the rollout continued
    in pieces on the same handle: after the reset the rows the kernel skips hold +0.0 bit for bit (pose, and velocity --
    checked on the lanes not yet in the scene), and the final state, metrics and events equal the oracle's."""
    import scenario_gym_amd._lib as L
    from oracle import check
    from scenario_gym_amd import synthetic

    R, E, steps, dt = 72, 64, 300, 1 / 30
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=L.KIND_AGENT_PID, static_frac=0.1, vanish_frac=0.5, extent=30.0)
    assert not packed.knots[:, [3, 5, 6]].any()  # planar
    monkeypatch.setenv("SG_QUEUE", queue)
    eng = sga.RolloutEngine(R, E, timestep=dt, persist=persist, terminal_conditions=["max_length"], event_capacity=64)
    eng.set_tuning(tab_min_steps=1, chunk_steps=16)
    eng.upload(packed)
    st = eng.state(raw=True)
    absent = ~st["present"].astype(bool)
    assert persist or absent.sum() > R * E // 4  # (persist: replay entities are in the scene from the reset on, state.py:99, entity/batch.py)
    for k in ("poses", "vels"):
        assert not np.ascontiguousarray(st[k][..., [2, 4, 5]]).view(np.uint64).any(), k  # +0.0: no sign bit, no NaN
    eng.rollout(150)
    assert persist or (eng.state()["present"].astype(bool) & absent).sum() > R  # entities joined the scenes on the way
    eng.rollout(90)
    eng.rollout_async(130, do_reset=False)
    eng.rollout_async(steps, do_reset=False)  # (max_length ends every scenario before the count runs out)
    eng.synchronize()
    ver = check.verify_engine(eng, packed, dt, steps, K=R, event_cap=64, persist=persist)
    assert ver["equal"], ver["mismatches"]
    st = eng.state(raw=True)
    assert not np.ascontiguousarray(st["poses"][..., [2, 4, 5]]).view(np.uint64).any()
    eng.close()


@pytest.mark.parametrize("zpr", [False, True])
def test_table_kernels_on_a_clock_that_stands_still(sga, oracle, zpr):
    """A timestep below half an ulp of the clock (1e-300 on scenarios that start at t = 1): t + timestep == t, next_t - t == 0
    in every step, so every velocity is 0 / 0 = NaN (state.py:148-152 divides by the clock difference) -- the planar table
    kernel must not take its "z / pitch / roll velocity rows hold +0" shortcut there (`flat` requires dt > 0), nor the
    general one its shared reciprocal.  Both kernels against the oracle."""
    import scenario_gym_amd._lib as L
    from oracle import check
    from scenario_gym_amd import synthetic

    R, E, steps = 16, 64, 12
    packed = synthetic.make_batch(R, E, n_steps=300, ego_kind=L.KIND_AGENT_PID, static_frac=0.1, vanish_frac=0.3, extent=30.0)
    if zpr:
        packed.knots[:, 3] = np.random.default_rng(3).normal(0.0, 1.0, len(packed.knots))
    packed.knots[:, 0] += 1.0
    packed.t0 += 1.0
    packed.length += 1.0
    dt = 1e-300
    eng = sga.RolloutEngine(R, E, timestep=dt, terminal_conditions=["max_length"], event_capacity=64)
    eng.set_tuning(tab_min_steps=1, chunk_steps=4)
    eng.upload(packed)
    eng.rollout(steps)
    ver = check.verify_engine(eng, packed, dt, steps, K=R, event_cap=64)
    assert ver["equal"], ver["mismatches"]
    st = eng.state()
    assert (st["n_steps"] == steps).all() and (st["t"] == 1.0).all() and np.isnan(st["vels"][st["present"].astype(bool)]).all()
    eng.close()


@pytest.mark.parametrize("ego", ["replay", "pid"])
def test_rss_records_start_anew_on_every_upload(sga, oracle, ego):
    """A sweep reuses one handle for batch after batch (BatchedScenarioGym.set_packed): the RSS records and the line-test queue
    stay allocated across sg_upload -- freeing and re-allocating the queue (GiBs) stalled every tenth or so upload for a
    second -- but their CONTENTS belong to a batch: after the next upload sg_rss_read refuses until the callback has run, and
    the second batch's flags, codes and safe distances equal those of a fresh handle and the oracle's."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic
    from scenario_gym_amd.packing import unpack_scenario

    R, E, steps = 24, 40, 60
    kind = dict(replay=L.KIND_AGENT_REPLAY, pid=L.KIND_AGENT_PID)[ego]
    a = synthetic.make_batch(R, E, n_steps=steps, ego_kind=kind, extent=20.0, vanish_frac=0.3)
    b = synthetic.make_batch(R, E, n_steps=steps, ego_kind=kind, extent=16.0, vanish_frac=0.2, seed=99)
    eng = sga.RolloutEngine(R, E)
    eng.set_rss(True)
    eng.upload(a)
    eng.rollout(steps)
    first = eng.rss()
    eng.set_rss(False)
    eng.upload(b)
    with pytest.raises(RuntimeError, match="has not run on this batch"):
        eng.rss()
    eng.set_rss(True)
    eng.upload(b)
    eng.rollout(steps)
    again = eng.rss()
    eng.close()
    fresh = sga.RolloutEngine(R, E)
    fresh.set_rss(True)
    fresh.upload(b)
    fresh.rollout(steps)
    want = fresh.rss()
    fresh.close()
    for x, y in zip(again, want):
        assert np.array_equal(x, y, equal_nan=True)
    assert not all(np.array_equal(x, y, equal_nan=True) for x, y in zip(first, want))
    for r in range(0, R, 5):
        s = unpack_scenario(b, r)
        o = oracle.rollout(s["knot_off"], s["knots"], s["bbox"], s["etype"], s["kind"], s["ego"], s["t0"], s["length"], 1 / 30,
                           ctrl=s["ctrl"], max_steps=steps)
        w = oracle.rss_rollout(o, s["bbox"], s["ego"])
        assert np.array_equal(again[2][r], w["code"][-1]) and np.array_equal(again[3][r], w["safe"][-1], equal_nan=True), r


def test_launch_stats_of_both_schedules(sga, monkeypatch):
    """sg_last_launch_stats reports the rollout-kernel launches of the last call and the time during which at least one of
    them ran (the union of their intervals), sg_last_launch_gross_ms the plain sum of their durations.  The persistent table
    launch is ONE launch; the chunk launches (SG_QUEUE=0) are several on one stream: union and sum agree in both, and both stay
    within the device time of the whole call.  (Counts only: nothing here compares speeds.)"""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    R, E, steps = 1024, 64, 400
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=L.KIND_AGENT_PID)
    out = {}
    for queue in ("0", "1"):
        monkeypatch.setenv("SG_QUEUE", queue)
        eng = sga.RolloutEngine(R, E)
        eng.set_slicing(False)
        eng.upload(packed)
        eng.rollout(steps)
        eng.rollout(steps)
        n, net = eng.last_launch_stats()
        out[queue] = (n, net, eng.last_launch_gross_ms(), eng.last_kernel_ms(), eng.schedule_info())
        eng.close()
    n0, net0, gross0, call0, info0 = out["0"]
    n1, net1, gross1, call1, info1 = out["1"]
    assert info0["schedule"] == SCHED_CHUNKS and info1["schedule"] == SCHED_QUEUE
    assert n0 > 1 and n1 == 1 and info1["launches"] == 1 and info1["grid"] == info1["blocks"] + info1["ctl_waves"]
    for net, gross, call in ((net0, gross0, call0), (net1, gross1, call1)):
        assert abs(gross - net) <= 1e-3 * gross and 0.0 < net <= call * 1.001


def test_prepass_resumes_from_the_state_blocks(sga):
    """A rollout that ends early followed by forced steps (gym.step on a done scenario), and a rollout continued in
    pieces: the controller pre-pass restarts from the device state at every call."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    R, E, steps = 64, 16, 120
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=L.KIND_AGENT_PID, extent=20.0)
    out = []
    for tn in (INLINE, TABLE_SMALL_CHUNKS):
        eng = sga.RolloutEngine(R, E, terminal_conditions=["max_length", "ego_collision"], event_capacity=64)
        eng.set_tuning(**tn)
        eng.upload(packed)
        eng.rollout(60)
        eng.rollout_async(30, do_reset=False)   # continue the unfinished scenarios
        eng.step(25)                            # then force everybody
        st = eng.state()
        rows, ev = eng.metrics()
        eng.close()
        out.append((st, rows, ev))
    (a, ra, ea), (b, rb, eb) = out
    assert len(set(ra["n_steps"].tolist())) > 1
    for k in ("poses", "vels", "dists", "ctrl_state", "t"):
        assert bits_equal(a[k], b[k]), k
    assert np.array_equal(a["coll"], b["coll"]) and ra.tobytes() == rb.tobytes() and ea.tobytes() == eb.tobytes()


def test_trig32_error_bound(sga):
    """The hardware sin/cos of the broad phase stays inside SG_TRIG32_ERR (sgym_device.hpp), the bound its
    conservative margins are built on: dense sweep of one turn, random headings up to 1e5 rad, edge values."""
    eng = sga.RolloutEngine(1, 4)
    rng = np.random.default_rng(7)
    h = np.concatenate([
        np.linspace(-np.pi, np.pi, 1 << 22), rng.uniform(-40.0, 40.0, 1 << 21), rng.uniform(-1e5, 1e5, 1 << 21),
        np.arange(-64, 65) * (np.pi / 32), [0.0, -0.0, 1e-300, 99999.999, -99999.999, 1e5, 3e7, -1e12],
    ])
    s, c = eng.debug_trig32(h)
    err = max(np.abs(s - np.sin(h)).max(), np.abs(c - np.cos(h)).max())
    assert err <= 4.0e-6, err
    s, c = eng.debug_trig32(np.array([np.nan, np.inf]))
    assert np.isnan(s).all() and np.isnan(c).all()
    eng.close()


# --------------------------------------------------------------------------- the C boundary, full horizon
def test_plain_c_caller(tmp_path):
    """include/sgym.h driven from C11 (tests/c_abi/abi_smoke.c, built with gcc, no Python / C++ / torch in the process):
    upload, rollout, metrics, events, raw state view, error path."""
    import os
    import subprocess

    from conftest import ROOT

    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.join(ROOT, "scenario_gym_amd", "lib")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi", "abi_smoke.c"), "-o", exe, "-L", libdir, "-lsgym_hip", "-lm",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
    assert "collisions 1" in out.stdout and "collisions 0" in out.stdout


def test_full_horizon_matches_oracle(sga, oracle):
    """BASELINE config 3's full 10,000-step horizon (several pre-pass chunks, 1024 scenarios x 64 entities, PID ego):
    scattered scenarios bit-identical to the oracle at the end -- poses, velocities, distances, collision rows, ego
    metrics, every collision event -- and the whole batch idempotent."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    R, E, steps = 1024, 64, 10000
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=L.KIND_AGENT_PID)
    eng = sga.RolloutEngine(R, E, event_capacity=64)
    eng.upload(packed)
    eng.rollout(steps)
    st = eng.state()
    rows, events = eng.metrics()
    n_launches, _ = eng.last_launch_stats()  # (a batch of 1024 blocks is time-sliced by default: several slice groups)
    # t accumulates in fp64: max_length (t + dt > length) fires one step before or at the nominal horizon
    assert n_launches >= 1 and (rows["n_steps"] >= steps - 1).all() and rows["done"].all()
    eng.rollout(steps)
    st2 = eng.state()
    rows2, _ = eng.metrics()
    eng.close()
    for k in ("poses", "vels", "dists", "ctrl_state"):
        assert bits_equal(st[k], st2[k]), k
    assert rows.tobytes() == rows2.tobytes()
    for r in (0, 511, 1023):
        o = _oracle_batch(oracle, packed, 1 / 30, steps, [r])[r]
        assert rows["n_steps"][r] == o["n_steps"] and rows["final_t"][r] == o["final_t"], r
        assert bits_equal(st["poses"][r], o["poses"][-1]) and bits_equal(st["vels"][r], o["vels"][-1]), r
        assert bits_equal(st["dists"][r], o["dists"][-1]) and np.array_equal(st["coll"][r], o["coll"][-1, :, 0]), r
        for k in ("ego_avg_speed", "ego_max_speed", "ego_distance_travelled"):
            assert rows[k][r] == o["metric_" + k], (r, k)
        ev = events[events["scenario"] == r]
        assert rows["n_collisions"][r] == o["n_events"]
        m = min(len(ev), 64)
        assert np.array_equal(ev["t"][:m], o["ev_t"][:m]) and np.array_equal(ev["other"][:m], o["ev_other"][:m]), r


def test_future_collision_batch_matches_oracle(sga, oracle):
    """sg_future_collision on a synthetic batch (512 x 24, dense scenes) at three state times and three horizons:
    every scenario's flag equals the oracle's; both outcomes occur."""
    from scenario_gym_amd import synthetic
    from scenario_gym_amd.packing import unpack_scenario

    R, E = 512, 24
    packed = synthetic.make_batch(R, E, n_steps=300, static_frac=0.15, vanish_frac=0.2, extent=25.0)
    eng = sga.RolloutEngine(R, E)
    eng.upload(packed)
    seen = set()
    for n_adv in (0, 70, 90):
        if n_adv:
            eng.step(n_adv)
        t = eng.state()["t"]
        for horizon, n in ((5.0, 10), (1.0, 10), (0.5, 4), (3.0, 70), (2.0, 1)):  # 70 samples: more than one LDS round
            got = eng.future_collision(horizon, n)
            for r in range(0, R, 3):
                s = unpack_scenario(packed, r)
                want = oracle.future_collision(s["knot_off"], s["knots"], s["bbox"], s["kind"], s["ego"], t[r], horizon, n)
                assert got[r] == want, (n_adv, horizon, r)
                seen.add(bool(want))
    eng.close()
    assert seen == {True, False}


def test_sensors_on_wide_scenarios_match_oracle(sga, oracle):
    """FutureCollisionDetector and the entity layer of RasterizedMapSensor on scenarios of 300 and 512 entities (one thread
    per entity slot: 512 threads per scenario) and of 1,100 (tile by tile): every flag / every cell equals the oracle's, after
    a reset and after 60 steps."""
    from scenario_gym_amd import synthetic
    from scenario_gym_amd.packing import unpack_scenario

    for R, E, extent in ((6, 300, 45.0), (4, 512, 60.0), (2, 1100, 85.0)):
        packed = synthetic.make_batch(R, E, n_steps=200, static_frac=0.15, vanish_frac=0.2, extent=extent)
        eng = sga.RolloutEngine(R, E)
        eng.upload(packed)
        cells = 0
        for n_adv in (0, 60):
            if n_adv:
                eng.step(n_adv)
            st = eng.state()
            fut = eng.future_collision(2.0, 10)
            ras = eng.raster_entities(30.0, 24.0, 25, 19)
            for r in range(R):
                s = unpack_scenario(packed, r)
                assert fut[r] == oracle.future_collision(s["knot_off"], s["knots"], s["bbox"], s["kind"], s["ego"], st["t"][r], 2.0, 10), (E, r)
                want = oracle.raster_entities(st["poses"][r, :len(s["bbox"])], s["bbox"], s["ego"], 30.0, 24.0, 25, 19)
                assert np.array_equal(ras[r], want), (E, n_adv, r)
                cells += int(want.sum())
        eng.close()
        assert cells > 100


@pytest.mark.parametrize("R,E,extent,n_adv2", [(256, 40, 20.0, 85), (5, 300, 45.0, 40), (3, 1300, 90.0, 12)])
def test_raster_entities_batch_matches_oracle(sga, oracle, R, E, extent, n_adv2):
    """sg_raster_entities on a synthetic batch (256 x 40, dense scenes, vanishing entities; scenarios of 300 and of 1,300
    entities: the kernel goes over the entities tile by tile) at two state times and two grids (one not square): every cell
    of every scenario equals the oracle's; sg_raster_map's entity layer is the same grid."""
    from scenario_gym_amd import synthetic
    from scenario_gym_amd.packing import unpack_scenario

    packed = synthetic.make_batch(R, E, n_steps=300, static_frac=0.15, vanish_frac=0.2, extent=extent)
    eng = sga.RolloutEngine(R, E)
    eng.upload(packed)
    total = 0
    for n_adv in (0, n_adv2):
        if n_adv:
            eng.step(n_adv)
        st = eng.state()
        for (w, h, nw, nh) in ((20.0, 20.0, 20, 20), (40.0, 16.0, 33, 12)):
            got = eng.raster_entities(w, h, nw, nh)
            assert got.shape == (R, nh, nw)
            for r in range(0, R, 2 if R > 8 else 1):
                s = unpack_scenario(packed, r)
                want = oracle.raster_entities(st["poses"][r, :len(s["bbox"])], s["bbox"], s["ego"], w, h, nw, nh)
                assert np.array_equal(got[r], want), (n_adv, w, r, int((got[r] != want).sum()))
                total += int(want.sum())
            if E > 64:
                assert np.array_equal(eng.raster_map([0], w, h, nw, nh)[:, 0], got.astype(bool))
    eng.close()
    assert total > (2000 if R > 8 else 200)


def test_lattice_scenes_touching_boxes_every_step(sga, oracle):
    """Adversarial geometry for the fp32 filter / exact path split: axis-aligned 2 x 4 boxes on an integer lattice, moving
    by exact lattice fractions, so that boxes touch along whole edges, at single corners, coincide exactly or slide past
    at zero gap for many steps (closed-set semantics: touching collides; identical geometries never list each other).
    Collision rows of every scenario are compared with the oracle after EVERY step."""
    from scenario_gym_amd.packing import default_kinds, pack_arrays

    rng = np.random.default_rng(5)
    R, E, steps, dt = 48, 12, 40, 0.25
    scs = []
    for r in range(R):
        knots, off = [], [0]
        for e in range(E):
            x0, y0 = rng.integers(-6, 7) * 2.0, rng.integers(-3, 4) * 2.0       # box length 4 along x, width 2 along y
            vx, vy = rng.choice([0.0, 1.0, -1.0, 2.0]), rng.choice([0.0, 0.0, 1.0, -1.0])
            h = rng.choice([0.0, 0.0, 0.0, np.pi])                                # pi: the same box up to rounding
            if rng.random() < 0.25:
                knots.append([0.0, x0, y0, 0, h, 0, 0])                          # static
                off.append(off[-1] + 1)
            else:
                T = steps * dt
                knots += [[0.0, x0, y0, 0, h, 0, 0], [T, x0 + vx * T, y0 + vy * T, 0, h, 0, 0]]
                off.append(off[-1] + 2)
        scs.append(dict(knot_off=np.array(off), knots=np.array(knots, np.float64), bbox=np.tile([2.0, 4.0, 0.0, 0.0], (E, 1)),
                        etype=np.full(E, 2, np.int32), ego=0, t0=0.0, length=steps * dt))
    packed = pack_arrays(scs)
    eng = sga.RolloutEngine(R, E, timestep=dt, event_capacity=256)
    eng.upload(packed)
    ref = [oracle.rollout(s["knot_off"], s["knots"], s["bbox"], s["etype"], default_kinds(E, 0), 0, 0.0, s["length"], dt,
                          max_steps=steps, event_cap=512) for s in scs]
    n_touch = 0
    for k in range(1, steps + 1):
        eng.step(1)
        st = eng.state()
        for r in range(R):
            if k <= ref[r]["n_steps"]:
                assert np.array_equal(st["coll"][r], ref[r]["coll"][k, :, 0]), (r, k)
                n_touch += int(np.count_nonzero(st["coll"][r]))
    rows, events = eng.metrics()
    eng.close()
    for r in range(R):
        ev = events[events["scenario"] == r]
        n = min(len(ev), 256)
        assert rows["n_collisions"][r] >= ref[r]["n_events"] or ref[r]["n_steps"] < steps
        if ref[r]["n_steps"] == steps:
            assert rows["n_collisions"][r] == ref[r]["n_events"] and np.array_equal(ev["t"][:n], ref[r]["ev_t"][:n]), r
    assert n_touch > 2000


def _random_configs(n, seed=2024):
    rng = np.random.default_rng(seed)
    zrng = np.random.default_rng([seed, 3])  # (its own stream: the configurations of earlier rounds keep their draws)
    wrng = np.random.default_rng([seed, 4])
    widths = [1, 3, 4, 5, 8, 9, 16, 17, 31, 33, 63, 64, 65, 100, 128, 129, 200, 256, 257, 400, 512]
    out = []
    for k in range(n):
        E = int(widths[k % len(widths)])
        out.append(dict(
            E=E, R=int(rng.integers(3, 20 if E <= 64 else (8 if E <= 256 else 4))), steps=int(rng.integers(20, 110)),
            dt=float(rng.choice([1 / 30, 0.05, 0.1, 0.013])), persist=bool(rng.integers(0, 2)),
            ego=str(rng.choice(["replay", "pid", "vehicle"])),
            terminal=[["max_length"], ["max_length", "collision"], ["max_length", "ego_collision"]][int(rng.integers(0, 3))],
            static=float(rng.choice([0.0, 0.15, 0.5])), vanish=float(rng.choice([0.0, 0.2, 0.6])),
            extent=float(rng.choice([8.0, 25.0, 60.0])), knots=int(rng.choice([6, 9, 40])), seed=int(rng.integers(1, 1 << 30)),
            chunk=int(rng.choice([5, 16, 1024])),
            zpr=bool(zrng.integers(0, 2))))  # knots with z / pitch / roll: the general table kernel instead of the planar one
        if wrng.integers(0, 12) == 0:        # one in twelve: more than 512 entities (the multi-kernel step, sgym_wide.hpp)
            out[-1].update(E=int(wrng.choice([513, 600, 777, 1100])), R=int(wrng.integers(1, 4)), steps=min(out[-1]["steps"], 45),
                           extent=float(wrng.choice([25.0, 60.0, 120.0])))
    return out


# SG_FUZZ_N / SG_FUZZ_SEED widen the sweep for soak runs (tools/fuzz.sh); the default is the fixed set of 36
@pytest.mark.parametrize("cfg", _random_configs(int(os.environ.get("SG_FUZZ_N", "36")), int(os.environ.get("SG_FUZZ_SEED", "2024"))), ids=lambda c: f"E{c['E']}-{c['ego']}-{'p' if c['persist'] else 'n'}-{len(c['terminal'])}{c['terminal'][-1][0]}")
def test_randomized_configurations_match_oracle(sga, oracle, cfg):
    """Differential test over the configuration space: every tile width across the wavefront boundaries (1 ... 256
    entities), both persist modes, every terminal condition, replay / PID / external-action egos, static-heavy and
    vanishing-heavy scenes, sparse and dense, 6 ... 40 knots, four time steps, short pre-pass chunks: recorded poses of
    every step, final velocities / distances / collision rows / controller state, metrics and events equal the oracle's."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic
    from scenario_gym_amd.engine import TERMINAL_BITS

    kind = dict(replay=L.KIND_AGENT_REPLAY, pid=L.KIND_AGENT_PID, vehicle=L.KIND_AGENT_VEHICLE)[cfg["ego"]]
    R, E, steps, dt = cfg["R"], cfg["E"], cfg["steps"], cfg["dt"]
    packed = synthetic.make_batch(R, E, n_steps=steps, timestep=dt, n_knots=cfg["knots"], ego_kind=kind,
                                  static_frac=cfg["static"], vanish_frac=cfg["vanish"], extent=cfg["extent"], seed=cfg["seed"])
    if cfg["zpr"]:
        zr = np.random.default_rng([cfg["seed"], 11])
        packed.knots[:, 3] = zr.uniform(-2.0, 2.0, len(packed.knots))
        packed.knots[:, 5] = zr.uniform(-0.2, 0.2, len(packed.knots))
        packed.knots[:, 6] = np.where(zr.random(len(packed.knots)) < 0.5, 0.0, zr.uniform(-0.1, 0.1, len(packed.knots)))
    force = cfg["ego"] == "vehicle"
    acts = synthetic.make_actions(steps, R, seed=cfg["seed"]) if force else None
    st, rows, events, t, poses = _engine_run(sga, packed, dt, steps, persist=cfg["persist"], terminal=cfg["terminal"],
                                             actions=acts, ev_cap=256, tuning=dict(tab_min_steps=8, chunk_steps=cfg["chunk"]))
    mask = sum(TERMINAL_BITS[c] for c in cfg["terminal"])
    for r in range(R):
        kw = dict(actions=acts[:, r], force_steps=True) if force else {}
        o = _oracle_one(oracle, packed, r, dt, steps, persist=cfg["persist"], terminal_mask=mask, event_cap=512, **kw)
        n = o["n_steps"]
        assert rows["n_steps"][r] == n and rows["final_t"][r] == o["final_t"], r
        assert bool(rows["done"][r]) == o["is_done"], r
        assert bits_equal(t[: n + 1, r], o["t"]) and bits_equal(poses[: n + 1, r], o["poses"]), r
        assert bits_equal(st["vels"][r], o["vels"][-1]) and bits_equal(st["dists"][r], o["dists"][-1]), r
        assert np.array_equal(_dense_words(st["coll"][r], E), oracle.coll_to_dense(o["coll"], E)[-1]), r
        for k in ("ego_avg_speed", "ego_max_speed", "ego_distance_travelled"):
            a, b = rows[k][r], o["metric_" + k]
            assert a == b or (np.isnan(a) and np.isnan(b)), (r, k, a, b)
        ev = events[events["scenario"] == r]
        assert rows["n_collisions"][r] == o["n_events"], r
        m = min(len(ev), 256)
        assert np.array_equal(ev["t"][:m], o["ev_t"][:m]) and np.array_equal(ev["other"][:m], o["ev_other"][:m]), r
        # collision classes: controlled egos take their event pose from the controller table (two-kernel path) or from the
        # rollout kernel (short chunks run the controllers in-kernel)
        assert np.array_equal(ev["type"][:m], o["ev_type"][:m]), (r, ev["type"][:m], o["ev_type"][:m])
    if cfg["ego"] in ("replay", "pid") and E <= 64:  # the same batch through the time-sliced path: the same final results
        eng = sga.RolloutEngine(R, E, timestep=dt, persist=cfg["persist"], terminal_conditions=cfg["terminal"], event_capacity=256)
        eng.set_slicing("always")
        eng.upload(packed)
        eng.rollout(steps)
        st2 = eng.state()
        rows2, events2 = eng.metrics()
        eng.close()
        for k in ("poses", "vels", "dists", "t", "prev_t", "ctrl_state"):
            assert bits_equal(st[k], st2[k]), k
        assert np.array_equal(st["coll"], st2["coll"]) and np.array_equal(st["present"], st2["present"])
        assert np.array_equal(rows, rows2) and np.array_equal(events, events2)


@pytest.mark.parametrize("E,side", [(12, 8.0), (40, 12.0), (150, 22.0)])
def test_mixed_pedestrians_and_vehicles_match_oracle(sga, oracle, E, side):
    """One scene with every kind: a PID-controlled car (ego) and a replayed car drive through a social-force crowd (the
    pedestrian variant runs the vehicle controllers in-kernel, parameters from the static rows): poses of every step,
    forces, collision rows, events, ego metrics bit-identical to the oracle."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    R, steps, dt = 6, 90, 1 / 30
    packed = synthetic.make_crowd(R, E, n_steps=steps, side=side)
    T = steps * dt
    for r in range(R):
        for slot, (kind, y, v) in enumerate([(L.KIND_AGENT_PID, -1.0, 4.0), (L.KIND_REPLAY, 2.5, -3.0)]):
            i = r * E + slot
            a = int(packed.knot_off[i])
            assert packed.knot_off[i + 1] - a == 2
            x0 = -np.sign(v) * side / 2
            packed.knots[a] = [0.0, x0, y, 0.0, 0.0 if v > 0 else np.pi, 0.0, 0.0]
            packed.knots[a + 1] = [T, x0 + v * T, y, 0.0, 0.0 if v > 0 else np.pi, 0.0, 0.0]
            packed.kind[i], packed.etype[i] = kind, 0
            packed.bbox[i] = synthetic.CAR1_BBOX
            packed.ctrl[i] = sga.engine.DEFAULT_CTRL
    st, rows, events, t, poses = _engine_run(sga, packed, dt, steps, ev_cap=512)
    for r in range(R):
        o = _oracle_one(oracle, packed, r, dt, steps, event_cap=1024)
        n = o["n_steps"]
        assert rows["n_steps"][r] == n, r
        assert bits_equal(poses[: n + 1, r], o["poses"]), r
        assert bits_equal(st["vels"][r], o["vels"][-1]) and bits_equal(st["dists"][r], o["dists"][-1]), r
        ped = packed.kind[r * E:(r + 1) * E] == L.KIND_AGENT_PEDESTRIAN  # the oracle's extra columns are per kind
        assert bits_equal(st["force"][r][ped], o["extra"][-1, ped, 2:]), r
        assert np.array_equal(_dense_words(st["coll"][r], E), oracle.coll_to_dense(o["coll"], E)[-1]), r
        for k in ("ego_avg_speed", "ego_max_speed", "ego_distance_travelled"):
            assert rows[k][r] == o["metric_" + k], (r, k)
        ev = events[events["scenario"] == r]
        assert rows["n_collisions"][r] == o["n_events"]
        m = min(len(ev), 512)
        assert np.array_equal(ev["t"][:m], o["ev_t"][:m]) and np.array_equal(ev["other"][:m], o["ev_other"][:m]), r
    assert rows["n_collisions"].sum() > 0  # the car does plough through the crowd


@pytest.mark.parametrize("E,side", [(12, 8.0), (150, 22.0), (300, 30.0), (600, 40.0)])
def test_ego_off_road_with_pedestrian_agents(sga, oracle, E, side):
    """terminal_conditions = max_length + ego_off_road on scenes WITH pedestrian agents (a PID car -- entity 0 -- drives through
    a social-force crowd on a strip of road that ends before its trajectory does): the pedestrian rollout variants do not carry
    the condition, it runs as a launch of its own behind every step; every scenario stops at the oracle's step with the
    oracle's state -- one wavefront per tile, four, eight, and the multi-kernel step."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic
    from scenario_gym_amd.packing import unpack_scenario

    R, steps, dt = 5, 90, 1 / 30
    packed = synthetic.make_crowd(R, E, n_steps=steps, side=side)
    T = steps * dt
    for r in range(R):
        i = r * E
        a = int(packed.knot_off[i])
        packed.knots[a] = [0.0, -side / 2, -1.0 + 0.3 * r, 0.0, 0.0, 0.0, 0.0]
        packed.knots[a + 1] = [T, -side / 2 + 5.0 * T, -1.0 + 0.3 * r, 0.0, 0.0, 0.0, 0.0]
        packed.kind[i], packed.etype[i] = L.KIND_AGENT_PID, 0
        packed.bbox[i] = synthetic.CAR1_BBOX
        packed.ctrl[i] = sga.engine.DEFAULT_CTRL
    # a strip of road from the left edge to a different x per network; scenario 4 has no network at all (off after one step)
    nets = [dict(ring_off=[0, 1], vert_off=[0, 4], verts=np.array([[-side, -6.0], [x1, -6.0], [x1, 6.0], [-side, 6.0]]), layers=[1 | 16])
            for x1 in (-side / 2 + 4.0, -side / 2 + 9.0)]
    net_of = np.array([0, 1, 0, 1, -1], np.int32)
    eng = sga.RolloutEngine(R, E, timestep=dt, terminal_conditions=["max_length", "ego_off_road"], event_capacity=256)
    eng.upload(packed)
    eng.set_road_networks(nets, net_of)
    eng.rollout(steps)
    st = eng.state()
    rows, events = eng.metrics()
    eng.close()
    for r in range(R):
        s = unpack_scenario(packed, r)
        o = oracle.rollout(s["knot_off"], s["knots"], s["bbox"], s["etype"], s["kind"], s["ego"], s["t0"], s["length"], dt,
                           ctrl=s["ctrl"], route_off=s["route_off"], routes=s["routes"], max_steps=steps, event_cap=256,
                           terminal_mask=1 | 8, road=nets[net_of[r]] if net_of[r] >= 0 else None)
        n = o["n_steps"]
        assert rows["n_steps"][r] == n and bool(rows["done"][r]) == o["is_done"], (r, rows["n_steps"][r], n)
        assert bits_equal(st["poses"][r], o["poses"][-1]) and bits_equal(st["dists"][r], o["dists"][-1]), r
        assert rows["n_collisions"][r] == o["n_events"], r
    assert 1 < rows["n_steps"][0] < rows["n_steps"][1] < steps and rows["n_steps"][4] == 1


@pytest.mark.parametrize("E,side,cluster", [(256, 40.0, 0), (256, 30.0, 150), (128, 16.0, 70), (64, 10.0, 30), (40, 6.0, 0)])
def test_pedestrian_pair_balancing_is_invisible(sga, oracle, monkeypatch, E, side, cluster):
    """The crowd kernel spreads the (pedestrian, neighbour) pairs of a wavefront evenly over its lanes; SG_PED_SERIAL=1
    keeps one pedestrian per lane.  Same bits either way and as the oracle -- also when a tight cluster inside a sparse
    crowd gives a few lanes far more neighbours than the hand-over list of a wavefront holds."""
    from scenario_gym_amd import synthetic

    R, steps = 6, 70
    packed = synthetic.make_crowd(R, E, n_steps=steps, side=side)
    if cluster:  # the first `cluster` pedestrians of every scenario start inside a 2.5 m square
        rng = np.random.default_rng(5)
        kn = packed.knots.reshape(R, E, 2, 7)
        rt = packed.routes.reshape(R, E, 2, 2)
        start = rng.uniform(-1.25, 1.25, (R, cluster, 2))
        kn[:, :cluster, :, 1:3] = start[:, :, None, :]
        rt[:, :cluster, 0] = start
    out = []
    for serial in ("1", "0"):
        monkeypatch.setenv("SG_PED_SERIAL", serial)
        eng = sga.RolloutEngine(R, E, record_capacity=steps + 1, event_capacity=2048)
        eng.upload(packed)
        eng.rollout(steps)
        out.append((eng.state(), eng.metrics(), eng.record(steps + 1)))
        eng.close()
    (sa, (ra, ea), (_, pa)), (sb, (rb, eb), (_, pb)) = out
    assert bits_equal(pa, pb) and np.array_equal(ea, eb) and np.array_equal(ra, rb)
    for k in ("poses", "vels", "dists", "force", "ctrl_state"):
        assert bits_equal(sa[k], sb[k]), k
    assert np.array_equal(sa["coll"], sb["coll"])
    for r in range(2):
        o = _oracle_one(oracle, packed, r, 1 / 30, steps)
        assert bits_equal(pb[: o["n_steps"] + 1, r], o["poses"]) and bits_equal(sb["force"][r], o["extra"][-1, :, 2:]), r


@pytest.mark.parametrize("E,side", [(256, 40.0), (150, 20.0), (64, 10.0)])
def test_crowd_kernel_equals_general_pedestrian_kernel(sga, monkeypatch, E, side):
    """All-pedestrian batches run rollout_kernel_crowd (lean pair arithmetic under per-step guards); SG_CROWD_KERNEL=0 sends
    them through the general pedestrian variant.  Same bits -- also far from the origin and with a coordinate
    for which the tile fails the per-step vote, so that the crowd kernel takes the guarded pair path."""
    from scenario_gym_amd import synthetic

    R, steps = 5, 60
    packed = synthetic.make_crowd(R, E, n_steps=steps, side=side)
    kn = packed.knots.reshape(R, E, 2, 7)
    rt = packed.routes.reshape(R, E, 2, 2)
    kn[R - 1, :, :, 1:3] += 3.0e7  # one scenario 30,000 km out: beyond the stripe cells of the broad phase
    rt[R - 1] += 3.0e7
    kn[R - 2, 0, :, 1] = 1e-250  # a coordinate below 2^-800: this tile fails crowd_sane at its first steps
    rt[R - 2, 0, 0, 0] = 1e-250
    out = []
    for crowd in ("0", "1"):
        monkeypatch.setenv("SG_CROWD_KERNEL", crowd)
        eng = sga.RolloutEngine(R, E, record_capacity=steps + 1, event_capacity=2048)
        eng.upload(packed)
        eng.rollout(steps)
        out.append((eng.state(), eng.metrics(), eng.record(steps + 1)))
        eng.close()
    (sa, (ra, ea), (_, pa)), (sb, (rb, eb), (_, pb)) = out
    assert bits_equal(pa, pb) and np.array_equal(ea, eb) and np.array_equal(ra, rb)
    for k in ("poses", "vels", "dists", "force", "ctrl_state"):
        assert bits_equal(sa[k], sb[k]), k
    assert np.array_equal(sa["coll"], sb["coll"])


@pytest.mark.parametrize("E,side,steps,chunk,noise,late,term", [
    (256, 40.0, 3600, 100, "off", False, ["max_length"]),
    (256, 40.0, 3000, 37, "device", True, ["max_length"]),
    (200, 30.0, 3000, 64, "stream", False, ["max_length"]),
    (256, 40.0, 2700, 1, "off", True, ["max_length"]),            # a chunk per step: every transition between the kernels
    (256, 14.0, 600, 100, "off", False, ["max_length", "ego_collision"]),
    (130, 25.0, 2500, 33, "device", False, ["max_length"]),
])
def test_long_crowd_rollouts_match_oracle(sga, oracle, E, side, steps, chunk, noise, late, term):
    """Long rollouts of all-pedestrian scenes through rollout_kernel_crowd, in calls of `chunk` steps (resumed calls: whatever
    the kernel carries from step to step has to survive the end of a launch) and in one call: same state, metric rows and
    events, bit for bit -- and both equal the oracle's.  Odd call lengths, late spawns, the three noise modes, a terminal
    condition that looks at collision rows (on a packed square); most pedestrians have arrived well before the end."""
    from oracle import check
    from scenario_gym_amd import synthetic

    R, dt = 6, 1 / 30
    packed = synthetic.make_crowd(R, E, n_steps=steps, side=side)
    rng = np.random.default_rng(E * 1000 + steps)
    if late:  # every seventh pedestrian joins the scene later (the spawn rule, scenario_gym.py:240-244)
        kn = packed.knots.reshape(R * E, 2, 7)
        kn[::7, 0, 0] = rng.uniform(0.1, 0.6, len(kn[::7])) * steps * dt
    kw, noise_of = {}, None
    if noise == "device":
        kw = dict(social_force=dict(std_lon=0.1, std_lat=0.05, noise="device", noise_seed=11))
        noise_of = lambda r: dict(mode="device", std_lon=0.1, std_lat=0.05, seed=11, scenario_index=r)  # noqa: E731
    elif noise == "stream":
        normals = np.random.RandomState(5).standard_normal((R, 2 * E * (steps + 1)))
        kw = dict(social_force=dict(std_lon=0.05, std_lat=0.1, noise="stream", normals=normals))
        noise_of = lambda r: dict(mode="stream", std_lon=0.05, std_lat=0.1, normals=normals[r])  # noqa: E731
    tm = 0
    for name in term:
        tm |= {"max_length": oracle.TERM_MAX_LENGTH, "ego_collision": oracle.TERM_EGO_COLLISION}[name]
    out = []
    for calls in (1, chunk):
        eng = sga.RolloutEngine(R, E, timestep=dt, terminal_conditions=term, event_capacity=256, **kw)
        eng.upload(packed)
        if calls == 1:
            eng.rollout(steps)
        else:
            for k0 in range(0, steps, max(chunk, 20)):  # (a call per step would be 3,000 launches: at least 20 steps a call)
                eng.rollout_async(min(max(chunk, 20), steps - k0), do_reset=(k0 == 0))
            eng.synchronize()
        out.append((eng.state(), eng.metrics()))
        if calls == 1:
            ver = check.verify_engine(eng, packed, dt, steps, K=3, event_cap=256, ped=True, noise_of=noise_of, terminal_mask=tm)
            assert ver["equal"], ver["mismatches"]
        eng.close()
    (sa, (ra, ea)), (sb, (rb, eb)) = out
    for k in ("poses", "vels", "dists", "force", "ctrl_state", "present"):
        if k in sa:
            assert bits_equal(sa[k], sb[k]), k
    assert np.array_equal(sa["coll"], sb["coll"])
    assert np.array_equal(ea, eb)
    for k in ra.dtype.names if hasattr(ra, "dtype") and ra.dtype.names else ra.keys():
        assert bits_equal(np.asarray(ra[k], np.float64), np.asarray(rb[k], np.float64)), k


def test_negative_zero_pose_delta_keeps_its_sign(sga):
    """velocity = delta / dt (state.py:226-233) for a delta of -0.0 is -0.0: the shared-reciprocal shortcut of the velocity
    division only takes +0 numerators (its fused correction step would turn -0 into +0)."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    R, E = 3, 5
    packed = synthetic.make_batch(R, E, n_steps=10, ego_kind=L.KIND_AGENT_REPLAY, extent=30.0)
    packed.kind[0::E] = L.KIND_AGENT_EXTERNAL  # the egos are run by the caller
    eng = sga.RolloutEngine(R, E)
    eng.upload(packed)
    poses = np.full((R, E, 6), np.nan)
    poses[:, 0] = [0.0, 3.0, 0.0, 0.5, 0.0, 0.0]
    eng.set_external_poses(poses)
    eng.step(1)
    poses[:, 0] = [-0.0, 3.0, 0.0, 0.5, 0.0, 0.0]  # x: +0.0 -> -0.0, delta = (-0.0) - (+0.0) = -0.0
    eng.set_external_poses(poses)
    eng.step(1)
    st = eng.state()
    eng.close()
    vx = st["vels"][:, 0, 0]
    assert (vx == 0.0).all() and np.signbit(vx).all(), vx
    assert np.signbit(st["poses"][:, 0, 0]).all()


def test_upload_from_page_locked_memory(sga):
    """PackedScenarios.pin() moves the knots into sg_host_alloc memory: sg_upload then sends them in pieces with the stage-1
    resample of each piece behind its copy -- the same device state as the upload from ordinary memory, and the array is
    released with its last view."""
    import gc

    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    R, E, steps = 37, 20, 60  # (37 scenarios: the pieces of the copy are not equal)
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=L.KIND_AGENT_PID, extent=15.0, vanish_frac=0.3)
    out = []
    for pinned in (False, True):
        if pinned:
            before = packed.knots.copy()
            packed.pin()
            assert np.array_equal(before, packed.knots) and not packed.knots.flags.owndata
        eng = sga.RolloutEngine(R, E, event_capacity=64)
        eng.upload(packed)
        s0 = eng.state()
        eng.rollout(steps)
        out.append((s0, eng.state(), eng.metrics()))
        eng.close()
    for k in ("poses", "vels", "dists"):
        assert bits_equal(out[0][0][k], out[1][0][k]) and bits_equal(out[0][1][k], out[1][1][k]), k
    assert np.array_equal(out[0][2][0], out[1][2][0]) and np.array_equal(out[0][2][1], out[1][2][1])
    a = L.pinned_empty((1000, 7))
    a[:] = 1.0
    v = a[10:20]
    del a
    gc.collect()
    assert v.sum() == 70.0  # the view keeps the allocation alive
    del v
    gc.collect()


# ---------------------------------------------------------------- road surfaces
ROAD_BITS =dict(driveable_surface=1, road=2, intersection=4, lane=8, walkable_surface=16, pavement=32, crossing=64)


def _road_arrays(g, net):
    return {k: g[f"net/{net}/{k}"] for k in ("ring_off", "vert_off", "verts", "layers")}


def test_raster_map_layers_match_reference_and_oracle(sga, oracle):
    """sg_raster_map with all eight RasterizedMapSensor layers: the four scenarios of roads.npz (one per shipped road
    network) as ONE batch sharing/alternating networks, at every recorded frame of the reference's rollouts and on both
    grids, equal the reference's maps cell for cell."""
    from scenario_gym_amd.packing import default_kinds, pack_arrays

    g = load_golden("roads")
    names = [str(n) for n in g["scenarios"]]
    scs = []
    for n in names:
        s = scenario_arrays(g, f"{n}/scenario")
        s["kind"] = default_kinds(len(s["bbox"]), s["ego"])
        scs.append(s)
    packed = pack_arrays(scs)
    nets = sorted({str(g[f"{n}/network"]) for n in names})
    eng = sga.RolloutEngine(packed.n_scenarios, packed.n_entities, timestep=0.1)
    eng.upload(packed)
    eng.set_road_networks([_road_arrays(g, k) for k in nets], [nets.index(str(g[f"{n}/network"])) for n in names])
    layers = [0] + [ROAD_BITS[str(x)] for x in g["layers"][1:]]
    frames = sorted({int(s) for n in names for s in g[f"{n}/map_steps"]})
    done, cells = 0, 0
    for f in frames:
        eng.step(f - done)
        done = f
        for c, (w, h, k) in enumerate(g["raster_cfg"]):
            got = eng.raster_map(layers, w, h, int(k), int(k))
            for r, n in enumerate(names):
                steps = list(g[f"{n}/map_steps"])
                if f in steps:
                    want = g[f"{n}/map{c}"][steps.index(f)].astype(bool)
                    assert np.array_equal(got[r], want), (n, f, c, int((got[r] != want).sum()))
                    cells += want.size
    eng.close()
    assert cells > 500000


def test_surface_raster_on_synthetic_polygons_matches_oracle(sga, oracle):
    """Random star-shaped and ring-shaped (with a hole) polygons, many networks, egos placed ON vertices and edges so
    that grid points hit boundaries exactly: device = oracle on every cell, every layer; scenarios without a network
    give empty surfaces."""
    from scenario_gym_amd import synthetic

    rng = np.random.default_rng(17)
    R, E = 64, 4
    packed = synthetic.make_batch(R, E, n_steps=50, extent=5.0)
    nets = []
    for n in range(5):
        rings, ring_off, layers = [], [0], []
        for q in range(12):
            c = rng.uniform(-40, 40, 2)
            m = int(rng.integers(3, 40))
            ang = np.sort(rng.uniform(0, 2 * np.pi, m))
            rad = rng.uniform(4, 25) * rng.uniform(0.4, 1.0, m)
            rings.append(np.round(c + rad[:, None] * np.stack([np.cos(ang), np.sin(ang)], 1), 2))  # 1 cm lattice
            if q % 3 == 0:  # a hole well inside
                rings.append(np.round(c + 0.2 * rad.min() * np.stack([np.cos(ang[::-1]), np.sin(ang[::-1])], 1), 2))
            ring_off.append(len(rings))
            layers.append(int(rng.integers(1, 128)))
        vert_off = np.concatenate([[0], np.cumsum([len(r) for r in rings])])
        nets.append(dict(ring_off=np.array(ring_off), vert_off=vert_off, verts=np.concatenate(rings), layers=np.array(layers)))
    net_of = rng.integers(-1, len(nets), R)
    # egos on the 1 cm lattice, heading 0 or pi/2, a 1 cm grid pitch: grid points fall on vertices and axis-parallel edges
    for r in range(R):
        a, b = packed.knot_off[r * E], packed.knot_off[r * E + 1]  # the ego's knots
        xy = np.round(rng.uniform(-30, 30, 2), 2)
        if r % 3 == 0 and net_of[r] >= 0:  # some egos exactly on a polygon vertex of their network
            v = nets[net_of[r]]["verts"]
            xy = v[rng.integers(0, len(v))]
        packed.knots[a:b, 1:3] = xy
        packed.knots[a:b, 4] = rng.choice([0.0, np.pi / 2])
    eng = sga.RolloutEngine(R, E)
    eng.upload(packed)
    eng.set_road_networks(nets, net_of)
    st = eng.state()
    layers = [1, 0, 2, 4, 8, 16, 32, 64]
    on = 0
    for (w, h, k) in ((0.4, 0.4, 41), (60.0, 60.0, 25)):
        got = eng.raster_map(layers, w, h, k, k)
        for r in range(R):
            net = None if net_of[r] < 0 else nets[net_of[r]]
            want = oracle.raster_map(st["poses"][r], packed.bbox[r * E:(r + 1) * E], 0, net, layers, w, h, k, k)
            assert np.array_equal(got[r], want), (r, w, int((got[r] != want).sum()))
            on += int(want[0].sum())
        assert not got[net_of < 0][:, [0, 2, 3, 4, 5, 6, 7]].any()
    eng.close()
    assert on > 1000


def test_ego_off_road_terminal_matches_reference_and_oracle(sga, oracle):
    """terminal_conditions = max_length + ego_off_road (state/state.py:397-407): the eight reference rollouts of roads.npz
    (recorded egos that stay on the road, copies drifting off it) as one batch over three road networks stop at the
    reference's step with the reference's ego pose; a scenario without a network is off the road after its first step."""
    from scenario_gym_amd.packing import default_kinds, pack_arrays

    g = load_golden("roads")
    keys = [f"{n}/{tag}" for n in g["scenarios"] for tag in ("onroad", "drift")]
    scs = []
    for k in keys + [keys[0]]:
        s = scenario_arrays(g, f"{k}/scenario")
        s["kind"] = default_kinds(len(s["bbox"]), s["ego"])
        scs.append(s)
    packed = pack_arrays(scs)
    nets = sorted({str(g[f"{n}/network"]) for n in g["scenarios"]})
    net_of = [nets.index(str(g[f"{k.split('/')[0]}/network"])) for k in keys] + [-1]
    eng = sga.RolloutEngine(packed.n_scenarios, packed.n_entities, timestep=0.1, terminal_conditions=["max_length", "ego_off_road"],
                            record_capacity=400)
    eng.upload(packed)
    eng.set_road_networks([_road_arrays(g, k) for k in nets], net_of)
    eng.rollout(390)
    st = eng.state()
    t, poses = eng.record(391)
    eng.close()
    early = 0
    for r, k in enumerate(keys):
        want = g[f"{k}/t"]
        n = len(want) - 1
        assert st["n_steps"][r] == n and st["done"][r], k
        assert bits_equal(t[: n + 1, r], want) and bits_equal(poses[n, r, scs[r]["ego"]], g[f"{k}/final_ego"]), k
        early += want[-1] + 0.2 < scs[r]["length"]
        o = oracle.rollout(scs[r]["knot_off"], scs[r]["knots"], scs[r]["bbox"], scs[r]["etype"], scs[r]["kind"], scs[r]["ego"],
                           scs[r]["t0"], scs[r]["length"], 0.1, terminal_mask=9, road=_road_arrays(g, nets[net_of[r]]))
        assert o["n_steps"] == n and bits_equal(poses[: n + 1, r, : len(scs[r]["bbox"])], o["poses"]), k
    assert early >= 2 and st["n_steps"][-1] == 1 and st["done"][-1]


@pytest.mark.parametrize("R,E", [(96, 8), (40, 3), (12, 100), (6, 200), (6, 300), (4, 512), (5, 700)])
def test_ego_off_road_with_controlled_egos_on_synthetic_roads(sga, oracle, R, E):
    """PID egos (entity 0) wandering over random polygon "roads": every scenario stops at the oracle's step -- cells
    wholly inside or outside answer from the grid, boundary cells through the exact test.  Tile widths 4, 8, the
    two-, four- and eight-wavefront scenarios and the multi-kernel step of scenarios beyond 512 entities."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic
    from scenario_gym_amd.packing import unpack_scenario

    rng = np.random.default_rng(23)
    steps = 400 if E <= 512 else 150
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=L.KIND_AGENT_PID, extent=30.0)
    nets = []
    for n in range(6):
        rings, layers = [], []
        for q in range(10):
            c = rng.uniform(-60, 60, 2)
            m = int(rng.integers(5, 60))
            ang = np.sort(rng.uniform(0, 2 * np.pi, m))
            rad = rng.uniform(20, 60) * rng.uniform(0.6, 1.0, m)
            rings.append(c + rad[:, None] * np.stack([np.cos(ang), np.sin(ang)], 1))
            layers.append(1 if q < 7 else 16)
        vert_off = np.concatenate([[0], np.cumsum([len(r) for r in rings])])
        nets.append(dict(ring_off=np.arange(len(rings) + 1), vert_off=vert_off, verts=np.concatenate(rings), layers=np.array(layers)))
    net_of = rng.integers(0, len(nets), R)
    eng = sga.RolloutEngine(R, E, terminal_conditions=["ego_off_road"])
    eng.upload(packed)
    eng.set_road_networks(nets, net_of)
    eng.rollout(steps)
    st = eng.state()
    eng.close()
    stopped = 0
    for r in range(R):
        s = unpack_scenario(packed, r)
        o = oracle.rollout(s["knot_off"], s["knots"], s["bbox"], s["etype"], s["kind"], s["ego"], s["t0"], s["length"], 1 / 30,
                           terminal_mask=8, ctrl=s["ctrl"], max_steps=steps, road=nets[net_of[r]])
        assert st["n_steps"][r] == o["n_steps"] and bool(st["done"][r]) == o["is_done"], r
        assert bits_equal(st["poses"][r, : len(s["bbox"])], o["poses"][-1]), r
        stopped += 1 < o["n_steps"] < steps
    assert stopped > R // 10


@pytest.mark.parametrize("R,E,crowd", [(40, 12, False), (9, 150, False), (24, 30, True), (6, 300, False), (5, 600, False), (4, 560, True)])
def test_masked_reset_equals_fresh_reset(sga, R, E, crowd):
    """sg_reset_scenarios: after some steps, the flagged scenarios are exactly in the state a full reset gives them and the
    others exactly where they were; stepping on from there equals stepping two engines that were treated wholesale.  (Also on
    eight-wavefront tiles and on the multi-kernel step of scenarios beyond 512 entities, vehicles and crowds.)"""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    packed = synthetic.make_crowd(R, E, n_steps=120, side=10.0 if E < 256 else 34.0) if crowd else \
        synthetic.make_batch(R, E, n_steps=120, ego_kind=L.KIND_AGENT_PID, extent=20.0 if E < 256 else 90.0)
    a, b, c = (sga.RolloutEngine(R, E, record_capacity=8) for _ in range(3))
    for e in (a, b, c):
        e.upload(packed)
    a.step(37)
    b.step(37)
    mask = np.random.default_rng(1).random(R) < 0.4
    a.reset_scenarios(mask)           # a: flagged scenarios restart
    # c stays at the reset state (fresh), b continues untouched
    keys = ("poses", "vels", "dists", "ctrl_state", "force", "present", "coll", "t", "prev_t", "n_steps", "done")
    for steps in (0, 25):
        if steps:
            for e in (a, b, c):
                e.step(steps)
        sa, sb, sc = a.state(), b.state(), c.state()
        for k in keys:
            assert bits_equal(sa[k][mask], sc[k][mask]), (k, steps)
            assert bits_equal(sa[k][~mask], sb[k][~mask]), (k, steps)
        ma, mb, mc = a.metrics()[0], b.metrics()[0], c.metrics()[0]
        for k in ("ego_avg_speed", "ego_max_speed", "ego_distance_travelled", "n_collisions"):
            assert bits_equal(ma[k][mask], mc[k][mask]) and bits_equal(ma[k][~mask], mb[k][~mask]), (k, steps)
    for e in (a, b, c):
        e.close()


def test_terminal_flags_match_state(sga, oracle):
    """sg_terminal_flags evaluates all four TERMINAL_CONDITIONS on the current state whatever the handle's mask."""
    from scenario_gym_amd import synthetic

    R, E = 64, 10
    packed = synthetic.make_batch(R, E, n_steps=60, extent=12.0)
    sq = np.array([[-8.0, -8.0], [8.0, -8.0], [8.0, 8.0], [-8.0, 8.0]])
    eng = sga.RolloutEngine(R, E)
    eng.upload(packed)
    assert (eng.terminal_flags() & 8).all()  # no road networks: off the road
    eng.set_road_networks([dict(ring_off=[0, 1], vert_off=[0, 4], verts=sq, layers=[1])], np.zeros(R, np.int32))
    seen = 0
    for n in (0, 20, 45, 10):
        eng.step(n)
        st = eng.state()
        fl = eng.terminal_flags()
        any_coll = np.array([(st["coll"][r][st["present"][r]] != 0).any() for r in range(R)])
        ego_coll = st["present"][:, 0] & (st["coll"][:, 0].reshape(R, -1) != 0).any(axis=1)
        x, y = st["poses"][:, 0, 0], st["poses"][:, 0, 1]
        on = st["present"][:, 0] & oracle.surface_contains(dict(ring_off=[0, 1], vert_off=[0, 4], verts=sq, layers=[1]), 1, x, y)
        length = packed.length
        assert np.array_equal((fl & 1) != 0, st["t"] + (st["t"] - st["prev_t"]) > length)
        assert np.array_equal((fl & 2) != 0, any_coll) and np.array_equal((fl & 4) != 0, ego_coll)
        assert np.array_equal((fl & 8) != 0, ~on)
        seen |= int(np.bitwise_or.reduce(fl))
    eng.close()
    assert seen == 15


@pytest.mark.parametrize("si", [0, 1])
def test_pedestrians_beside_a_building_match_reference(sga, oracle, si):
    """The boundary terms of the social force on the road network of examples/crowds.py (see the oracle test of the same
    name): reference closed loops <= 1e-8, bit-identical to the oracle; several copies of the scenario in one batch, some
    without the network (they walk as if the building were not there)."""
    from scenario_gym_amd.packing import unpack_scenario

    g = load_golden("ped_roads")
    one, E = _ped_packed(g, si)
    s = unpack_scenario(one, 0)
    from scenario_gym_amd.packing import pack_arrays

    R = 5
    packed = pack_arrays([s] * R, kinds=[s["kind"]] * R, ctrls=[s["ctrl"]] * R)
    net = {k: g[f"net/{k}"] for k in ("ring_off", "vert_off", "verts", "layers")}
    net_of = np.array([0, -1, 0, 0, -1], np.int32)
    for dtn, dt in (("dt30", 1 / 30), ("dt10", 0.1)):
        p = f"loop{si}/{dtn}"
        n = int(g[p + "/n_steps"])
        eng = sga.RolloutEngine(R, E, timestep=dt, record_capacity=n + 3, event_capacity=128)
        eng.upload(packed)
        eng.set_road_networks([net], net_of)
        eng.rollout(n + 2)
        st = eng.state()
        t, poses = eng.record(n + 3)
        eng.close()
        ref = g[p + "/poses"]
        o_road = oracle.rollout(s["knot_off"], s["knots"], s["bbox"], s["etype"], s["kind"], s["ego"], s["t0"], s["length"], dt,
                                ctrl=s["ctrl"], route_off=s["route_off"], routes=s["routes"], road=net)
        o_free = oracle.rollout(s["knot_off"], s["knots"], s["bbox"], s["etype"], s["kind"], s["ego"], s["t0"], s["length"], dt,
                                ctrl=s["ctrl"], route_off=s["route_off"], routes=s["routes"])
        for r in range(R):
            o = o_road if net_of[r] == 0 else o_free
            assert st["n_steps"][r] == o["n_steps"]
            assert bits_equal(poses[: n + 1, r], o["poses"]) and bits_equal(st["force"][r], o["extra"][-1, :, 2:]), (r, dtn)
            if net_of[r] == 0:
                assert np.array_equal(np.isnan(poses[: n + 1, r]), np.isnan(ref))
                assert np.nanmax(np.abs(poses[: n + 1, r] - ref)) < 1e-8
                ped = ~np.isnan(g[p + "/extra"][-1][:, 0])
                assert np.abs(st["force"][r, ped] - g[p + "/extra"][-1][ped, 2:]).max() < 1e-8
        assert np.nanmax(np.abs(o_road["poses"] - o_free["poses"])) > 0.1


@pytest.mark.parametrize("E,blocks,riders", [(100, 2, False), (256, 2, False), (200, 3, False), (180, 2, True),
                                             (60, 4, False), (120, 5, False)])  # (64 building edges: the staged table full; 100: not staged)
def test_crowd_kernels_take_road_networks(sga, oracle, monkeypatch, E, blocks, riders):
    """The reference's pedestrians need a road network to run at all (pedestrian/sensor.py:50-51) and feel its buildings
    (social_force.py:86-104, 190-211).  Batches that have one no longer leave the crowd kernels: the boundary terms are a
    phase behind the neighbour sums (rollout_kernel_crowd on 2 and 4 wavefronts, rollout_kernel_crowd_riders with a PID car).
    Same bits as the general pedestrian kernel (SG_CROWD_ROADS=0) and as the oracle, scenarios with and without a network in
    one batch; the network is felt (the rollout differs from the one without it)."""
    import scenario_gym_amd._lib as L
    from oracle import check
    from scenario_gym_amd import synthetic

    R, steps, dt = 6, 700, 1 / 30
    packed, net, net_of = synthetic.make_crowd_roads(R, E, n_steps=steps, side=30.0, blocks=blocks, building=30.0 / blocks - 5.0)
    net_of[2] = -1  # one scenario without a network
    if riders:  # a PID car on the centre street
        packed = _add_riders(packed_two_waypoints(packed), np.random.default_rng(4), 30.0, steps * dt, [("pid", 0, "car")])
    out = []
    for roads in ("1", "0", "off"):
        monkeypatch.setenv("SG_CROWD_ROADS", "0" if roads == "0" else "1")
        eng = sga.RolloutEngine(R, E, timestep=dt, terminal_conditions=["max_length"], event_capacity=256)
        eng.upload(packed)
        if roads != "off":
            eng.set_road_networks([net], net_of)
        eng.rollout(steps)
        out.append((eng.state(), eng.metrics()))
        if roads == "1":
            ver = check.verify_engine(eng, packed, dt, steps, K=R, event_cap=256, ped=True, road_of=lambda r: net if net_of[r] == 0 else None)
            assert ver["equal"], ver["mismatches"]
        eng.close()
    (sa, (ra, ea)), (sb, (rb, eb)), (sc, _) = out
    for k in ("poses", "vels", "dists", "force", "ctrl_state", "present"):
        assert bits_equal(sa[k], sb[k]), k
    assert np.array_equal(sa["coll"], sb["coll"]) and ra.tobytes() == rb.tobytes() and ea.tobytes() == eb.tobytes()
    assert bits_equal(sa["poses"][2], sc["poses"][2]) and np.abs(sa["poses"][0] - sc["poses"][0]).max() > 0.05


@pytest.mark.parametrize("E", [48, 130])
def test_boundary_force_filter_keeps_the_reference_tie_rules(sga, oracle, monkeypatch, E):
    """The nearest building edge of the boundary force (social_force.py:190-211, GEOS DistanceOp: the FIRST edge at the smallest
    rounded distance) is found by a filter on squared distances and the reference's sequence over its candidates
    (ped_boundary_terms, sgym_road.hpp).  Pedestrians placed where the filter has to keep several: on the centre line of a street
    (two walls equally far), on the diagonal of a building's corner (its two edges), in the middle of a crossing (four corners,
    eight edges), on a wall, on a corner, inside a building, far outside -- first step and the following ones equal the oracle
    bit for bit, through the crowd kernel (table staged in LDS) and the general pedestrian kernel (device memory)."""
    from oracle import check
    from scenario_gym_amd import synthetic

    R, steps, dt = 3, 40, 1 / 30
    packed, net, net_of = synthetic.make_crowd_roads(R, E, n_steps=steps, side=30.0, blocks=2, building=10.0)
    # buildings: [-12.5, -2.5] and [2.5, 12.5] in x and in y; streets along x = 0, y = 0, x = +-15, y = +-15
    special = [(0.0, -7.5), (0.0, 6.0), (-7.5, 0.0), (4.0, 0.0), (0.0, 0.0), (-1.0, -1.0), (-2.0, -2.0), (1.5, -1.5), (-2.5, -7.5),
               (2.5, 2.5), (-2.5, -2.5), (-7.5, -7.5), (-3.0, -7.5), (100.0, 50.0), (0.25, -7.5), (-14.0, -14.0), (-13.0, -13.0),
               (0.0, 1e-9), (1e-9, -7.5), (-2.5 - 1e-12, -2.5 - 1e-12), (0.5, 14.0), (-14.0, 0.0)]
    kn = packed.knots.reshape(R * E, 2, 7)
    rng = np.random.default_rng(8)
    for r in range(R):
        order = rng.permutation(E)[: len(special)]
        for e, (x, y) in zip(order, special):
            kn[r * E + e, :, 1], kn[r * E + e, :, 2] = x, y
    packed.validate()
    for roads in ("1", "0"):
        monkeypatch.setenv("SG_CROWD_ROADS", roads)
        eng = sga.RolloutEngine(R, E, timestep=dt, terminal_conditions=["max_length"], event_capacity=256)
        eng.upload(packed)
        eng.set_road_networks([net], net_of)
        for n in (1, steps):
            eng.rollout(n)
            ver = check.verify_engine(eng, packed, dt, n, K=R, event_cap=256, ped=True, road_of=lambda r: net)
            assert ver["equal"], (roads, n, ver["mismatches"])
        assert ("rollout_kernel_crowd" in eng.last_kernel()) == (roads == "1")
        eng.close()


@pytest.mark.parametrize("E", [40, 100, 256])
def test_crowd_routes_of_any_length_match_oracle(sga, oracle, E):
    """PedestrianAgent's goal update (LineString(route).project, pedestrian/agent.py:59-62) in the crowd kernels keeps the
    route's waypoints in registers when there are at most four (ped_goal_update2 / ped_goal_update_reg, sgym_agents.hpp) and
    walks device memory otherwise: routes of 2 ... 7 waypoints side by side in one batch, with a repeated waypoint (a
    segment of length zero) and a route whose pedestrian starts beyond its last waypoint, equal the oracle bit for bit."""
    from oracle import check
    from scenario_gym_amd import synthetic

    R, steps, dt = 3, 500, 1 / 30
    base = synthetic.make_crowd(R, E, n_steps=steps, timestep=dt, seed=12, side=24.0)
    rng = np.random.default_rng(5)
    two = base.routes.reshape(R * E, 2, 2)
    routes, off = [], [0]
    for i in range(R * E):
        n = int(rng.integers(2, 8))
        t = np.sort(rng.uniform(0.0, 1.0, n - 2))
        mid = two[i, 0] + t[:, None] * (two[i, 1] - two[i, 0]) + rng.normal(0, 1.5, (n - 2, 2))
        wp = np.concatenate([two[i, :1], mid, two[i, 1:]])
        if i % 11 == 3 and n > 2:
            wp[1] = wp[0]          # a segment of length zero
        if i % 13 == 5:
            wp = wp[::-1].copy()   # the pedestrian stands at the END of its route
        routes.append(wp)
        off.append(off[-1] + n)
    base.routes = np.concatenate(routes)
    base.route_off = np.array(off, np.int64)
    packed = base.validate()
    eng = sga.RolloutEngine(R, E, timestep=dt, terminal_conditions=["max_length"], event_capacity=256)
    eng.upload(packed)
    eng.rollout(steps)
    assert "rollout_kernel_crowd" in eng.last_kernel()
    ver = check.verify_engine(eng, packed, dt, steps, K=R, event_cap=256, ped=True)
    assert ver["equal"], ver["mismatches"]
    eng.close()


def packed_two_waypoints(packed):
    """_add_riders edits batches whose pedestrians have two waypoints each: keep the first and the last of every route."""
    R, E = packed.n_scenarios, packed.n_entities
    nw = int(packed.route_off[1] - packed.route_off[0])
    packed.routes = packed.routes.reshape(R * E, nw, 2)[:, [0, nw - 1]].reshape(-1, 2)
    packed.route_off = np.arange(R * E + 1, dtype=np.int64) * 2
    return packed.validate()


def test_collision_types_match_reference_code_and_oracle(sga, oracle):
    """CollisionMetric's classification (t_bone / head_on / rear_end / side_swipe / non_vehicle) for the 47 scenes of
    collision_types.npz as one ragged batch: times, hazards and types equal the output of the reference's own
    record_collision code; on a dense synthetic batch of vehicles (replay egos and PID egos) the types equal the oracle's."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic
    from scenario_gym_amd.packing import default_kinds, pack_arrays, unpack_scenario

    g = load_golden("collision_types")
    names = list(g["names"])
    scs = []
    for n in names:
        s = scenario_arrays(g, f"{n}/scenario")
        s["kind"] = default_kinds(len(s["bbox"]), s["ego"])
        scs.append(s)
    packed = pack_arrays(scs)
    eng = sga.RolloutEngine(packed.n_scenarios, packed.n_entities, timestep=0.05, event_capacity=16)
    eng.upload(packed)
    eng.rollout(200)
    rows, events = eng.metrics()
    points = eng.collision_points()
    eng.close()
    seen = set()
    for r, n in enumerate(names):
        sel = events["scenario"] == r
        ev = events[sel]
        assert np.array_equal(ev["t"], g[f"{n}/ev_t"]) and np.array_equal(ev["other"], g[f"{n}/ev_other"]), n
        assert np.array_equal(ev["type"], g[f"{n}/ev_type"]), (n, ev["type"], g[f"{n}/ev_type"])
        # CollisionPointMetric: the reference's (point, angle); box corners differ from numpy's by <= 1 ulp of sin / cos
        assert len(ev) == 0 or np.abs(points[sel] - g[f"{n}/ev_point"]).max() < 1e-10, n
        seen |= set(ev["type"].tolist())
    assert seen == {1, 2, 3, 4, 5}
    for ego_kind in (L.KIND_AGENT_REPLAY, L.KIND_AGENT_PID):
        R, E, steps = 96, 24, 300
        packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=ego_kind, extent=14.0)
        eng = sga.RolloutEngine(R, E, event_capacity=128)
        eng.upload(packed)
        eng.rollout(steps)
        rows, events = eng.metrics()
        points = eng.collision_points()
        eng.close()
        n_ev = 0
        for r in range(0, R, 3):
            s = unpack_scenario(packed, r)
            o = oracle.rollout(s["knot_off"], s["knots"], s["bbox"], s["etype"], s["kind"], s["ego"], s["t0"], s["length"], 1 / 30,
                               ctrl=s["ctrl"], max_steps=steps, record=False, event_cap=128)
            ev = events[events["scenario"] == r]
            assert np.array_equal(ev["t"], o["ev_t"]) and np.array_equal(ev["other"], o["ev_other"]), r
            assert np.array_equal(ev["type"], o["ev_type"]), (r, ev["type"], o["ev_type"])
            assert bits_equal(points[events["scenario"] == r], o["ev_point"]), r  # non-vehicle hazards included
            n_ev += len(ev)
        assert n_ev > 40


@pytest.mark.parametrize("path", ["table", "inline", "per_tick"])
def test_collisions_with_controlled_hazards_are_classified(sga, oracle, path):
    """CollisionMetric.record_collision (metrics/collision.py:81-203) when the hazard is itself a controlled agent (a second
    and a third PIDAgent per scenario, beside the PID ego): its pose at the event cannot be re-derived from a trajectory, it
    is saved beside the event -- from the controller table (two-kernel path), by the hazard's own lane (controllers in the
    rollout kernel), also when the rollout is driven tick by tick.  Types and collision points equal the oracle's."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic
    from scenario_gym_amd.packing import unpack_scenario

    R, E, steps = 64, 24, 260
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=L.KIND_AGENT_PID, extent=13.0, static_frac=0.05, vanish_frac=0.1)
    kind = packed.kind.reshape(R, E)
    if path == "table":
        # the two-kernel path serves two controlled lanes per wavefront of two 24-entity scenarios: every other scenario has
        # a PID ego and a PID hazard, the ones between them replay
        kind[0::2, 1] = L.KIND_AGENT_PID
        kind[1::2, 0] = L.KIND_AGENT_REPLAY
    else:
        kind[:, 1] = L.KIND_AGENT_PID      # entities 1 and 2 drive under their own PID controllers
        kind[:, 2] = L.KIND_AGENT_PID
    eng = sga.RolloutEngine(R, E, event_capacity=128)
    eng.upload(packed)
    if path == "per_tick":
        eng.reset()
        for _ in range(steps):
            eng.step(1)
    else:
        eng.rollout(steps)
        # the table path (here: the persistent launch) or the controllers inside the rollout kernel
        assert eng.schedule_info()["schedule"] == (SCHED_QUEUE if path == "table" else 0)
    rows, events = eng.metrics()
    points = eng.collision_points()
    eng.close()
    n_ctl_events = 0
    for r in range(R):
        s = unpack_scenario(packed, r)
        o = oracle.rollout(s["knot_off"], s["knots"], s["bbox"], s["etype"], s["kind"], s["ego"], s["t0"], s["length"], 1 / 30,
                           ctrl=s["ctrl"], max_steps=steps, record=False, event_cap=128, force_steps=path == "per_tick")
        sel = events["scenario"] == r
        ev = events[sel]
        assert np.array_equal(ev["t"], o["ev_t"]) and np.array_equal(ev["other"], o["ev_other"]), r
        assert np.array_equal(ev["type"], o["ev_type"]), (r, ev["type"], o["ev_type"])
        assert bits_equal(points[sel], o["ev_point"]), r
        n_ctl_events += int(np.isin(ev["other"], (1, 2)).sum())
    assert n_ctl_events > 10 and (events["type"] >= 1).all()


def test_device_group_equals_single_handle(sga, oracle):
    """sg_group_*: the batch cut into contiguous shards over several handles (here all on GPU 0, incl. uneven shards and a
    pedestrian-free ragged batch with PID egos): rollout + metrics + events equal the single-handle run row for row."""
    import ctypes as C

    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    R, E, steps = 50, 12, 150
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=L.KIND_AGENT_PID, extent=14.0)
    eng = sga.RolloutEngine(R, E, event_capacity=32)
    eng.upload(packed)
    eng.rollout(steps)
    rows1, ev1 = eng.metrics()
    lib = eng.lib
    for n_dev in (1, 3, 4):
        cfg = L.SgConfig(0, R, E, 0, L.TERM_MAX_LENGTH, 0, 32, 0, 1 / 30)
        devs = np.zeros(n_dev, np.int32)
        g = C.c_void_p()
        assert lib.sg_group_create(C.byref(cfg), n_dev, devs.ctypes.data, C.byref(g)) == 0
        assert lib.sg_group_size(g) == n_dev
        arrs = dict(kind=np.ascontiguousarray(packed.kind, np.int32), etype=np.ascontiguousarray(packed.etype, np.int32),
                    bbox=np.ascontiguousarray(packed.bbox), knot_off=np.ascontiguousarray(packed.knot_off, np.int64),
                    knots=np.ascontiguousarray(packed.knots), ctrl=np.ascontiguousarray(packed.ctrl),
                    ego=np.ascontiguousarray(packed.ego, np.int32), t0=np.ascontiguousarray(packed.t0),
                    length=np.ascontiguousarray(packed.length), route_off=None, routes=None)
        sc = L.SgScenarios(*[None if arrs[k] is None else arrs[k].ctypes.data for k, _ in L.SgScenarios._fields_])
        assert lib.sg_group_upload(g, C.byref(sc)) == 0, lib.sg_group_last_error(g)
        assert lib.sg_group_rollout(g, steps) == 0, lib.sg_group_last_error(g)
        rows = np.zeros(R, rows1.dtype)
        ev = np.zeros(4096, ev1.dtype)
        n_ev = C.c_int32()
        assert lib.sg_group_read_metrics(g, rows.ctypes.data, ev.ctypes.data, 4096, C.byref(n_ev)) == 0
        assert n_ev.value == len(ev1) and len(ev1) > 10
        for k in rows1.dtype.names:
            assert bits_equal(rows[k], rows1[k]), (n_dev, k)
        for k in ("t", "scenario", "other", "type"):
            assert np.array_equal(ev[: n_ev.value][k], ev1[k]), (n_dev, k)
        assert lib.sg_group_destroy(g) == 0
    eng.close()


def _group_against_one_handle(sga, devs):
    """sg_group over the devices `devs` (one handle each, contiguous shards of 4096 x 64) against ONE handle with the whole batch
    on device 0: metric rows and events equal row for row; the single handle's first / last scenarios equal the oracle."""
    import ctypes as C

    import scenario_gym_amd._lib as L
    from oracle import check
    from scenario_gym_amd import synthetic

    R, E, steps = 4096, 64, 1000
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=L.KIND_AGENT_PID)
    eng = sga.RolloutEngine(R, E, event_capacity=32)
    eng.upload(packed)
    eng.rollout(steps)
    rows1, ev1 = eng.metrics()
    ver = check.verify_engine(eng, packed, 1 / 30, steps, K=4, event_cap=32, threads=8)
    assert ver["equal"], ver["mismatches"]
    lib = eng.lib
    cfg = L.SgConfig(0, R, E, 0, L.TERM_MAX_LENGTH, 0, 32, 0, 1 / 30)
    devs = np.asarray(devs, np.int32)
    g = C.c_void_p()
    assert lib.sg_group_create(C.byref(cfg), len(devs), devs.ctypes.data, C.byref(g)) == 0
    arrs = dict(kind=np.ascontiguousarray(packed.kind, np.int32), etype=np.ascontiguousarray(packed.etype, np.int32),
                bbox=np.ascontiguousarray(packed.bbox), knot_off=np.ascontiguousarray(packed.knot_off, np.int64),
                knots=np.ascontiguousarray(packed.knots), ctrl=np.ascontiguousarray(packed.ctrl),
                ego=np.ascontiguousarray(packed.ego, np.int32), t0=np.ascontiguousarray(packed.t0),
                length=np.ascontiguousarray(packed.length), route_off=None, routes=None)
    sc = L.SgScenarios(*[None if arrs[k] is None else arrs[k].ctypes.data for k, _ in L.SgScenarios._fields_])
    assert lib.sg_group_upload(g, C.byref(sc)) == 0, lib.sg_group_last_error(g)
    assert lib.sg_group_rollout(g, steps) == 0, lib.sg_group_last_error(g)
    assert lib.sg_group_rollout(g, steps) == 0, lib.sg_group_last_error(g)  # (again: the handles' arrays are reused)
    rows = np.zeros(R, rows1.dtype)
    ev = np.zeros(1 << 17, ev1.dtype)
    n_ev = C.c_int32()
    assert lib.sg_group_read_metrics(g, rows.ctypes.data, ev.ctypes.data, len(ev), C.byref(n_ev)) == 0
    assert n_ev.value == len(ev1)
    for k in rows1.dtype.names:
        assert bits_equal(rows[k], rows1[k]), k
    for k in ("t", "scenario", "other", "type"):
        assert np.array_equal(ev[: n_ev.value][k], ev1[k]), k
    for i in range(len(devs)):  # every shard: time-sliced (<= 1024 blocks, schedule 0) or its own persistent launch -- never chunk launches
        info = (C.c_int32 * 8)()
        assert lib.sg_schedule_info(lib.sg_group_handle(g, i), info) == 0
        assert info[0] in (0, SCHED_QUEUE) and info[5] == R // len(devs), list(info)
    assert lib.sg_group_destroy(g) == 0
    eng.close()


def test_device_group_four_handles_at_the_timed_width(sga, oracle):
    """sg_group with FOUR handles on device 0, 1024 scenarios x 64 entities each, against ONE handle with all 4096 (the
    config-4 partitioning, on the one GPU there is).  The handles of a group launch before any of them is waited for: four
    persistent launches side by side on one device -- more wavefronts than it holds, so the later launches' wavefronts arrive as
    the earlier ones' leave (the role election and the work queues assume nothing about residency)."""
    _group_against_one_handle(sga, [0, 0, 0, 0])


def test_device_group_over_every_visible_device(sga, oracle):
    """The same on devices 0 .. N - 1 when more than one GPU is visible (config 4 as BASELINE.json states it; skipped on a
    one-GPU box): the first run on a multi-GPU node exercises sg_group across devices without anybody having to write a test."""
    import torch

    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("one visible device")
    n = max(d for d in (8, 4, 2) if d <= n)
    _group_against_one_handle(sga, list(range(n)))


def test_rss_distances_match_reference_and_oracle(sga, oracle):
    """sg_rss_update after every tick: the 46 scenarios of rss.npz as one ragged batch -- every record RSSDistances appended
    (per step, per entity), its safe distances and the two RSS metric flags equal the reference's; a dense synthetic batch
    with PID egos equals the oracle step by step."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic
    from scenario_gym_amd.packing import default_kinds, pack_arrays, unpack_scenario

    g = load_golden("rss")
    names = [str(n) for n in g["names"]]
    scs = []
    for n in names:
        s = scenario_arrays(g, f"{n}/scenario")
        s["kind"] = default_kinds(len(s["bbox"]), s["ego"])
        scs.append(s)
    packed = pack_arrays(scs)
    R, E = packed.n_scenarios, packed.n_entities
    steps = max(len(g[f"{n}/t"]) for n in names) - 1
    eng = sga.RolloutEngine(R, E, timestep=0.1)
    eng.upload(packed)
    eng.rss_update(reset=True)
    codes, safes = [eng.rss()[2]], [eng.rss()[3]]
    for k in range(steps):
        eng.lib.sg_rollout_async(eng.h, 1, 0)  # one step of the scenarios that are not done (as gym.rollout does)
        eng.rss_update()
        _, _, c, s = eng.rss()
        codes.append(c)
        safes.append(s)
    slong, slat, _, _ = eng.rss()
    n_steps = eng.state()["n_steps"]
    eng.close()
    codes, safes = np.array(codes), np.array(safes)
    for r, n in enumerate(names):
        want, ws = g[f"{n}/code"], g[f"{n}/safe"]
        T, Er = want.shape
        assert n_steps[r] == T - 1
        assert np.array_equal(codes[:T, r, :Er], want), (n, np.argwhere(codes[:T, r, :Er] != want)[:5])
        upd = want >= 0
        assert np.abs(safes[:T, r, :Er][upd] - ws[upd]).max() < 1e-9, n
        assert slong[r] == bool(g[f"{n}/safe_longitudinal"]) and slat[r] == bool(g[f"{n}/safe_lateral"]), n
    # controlled egos, dense traffic: against the oracle
    R, E, steps = 48, 16, 120
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=L.KIND_AGENT_PID, extent=18.0)
    eng = sga.RolloutEngine(R, E, timestep=1 / 30)
    eng.upload(packed)
    eng.rss_update(reset=True)
    codes = [eng.rss()[2]]
    for k in range(steps):
        eng.step(1)
        eng.rss_update()
        codes.append(eng.rss()[2])
    slong, slat, _, _ = eng.rss()
    eng.close()
    codes = np.array(codes)
    unsafe = 0
    for r in range(0, R, 2):
        s = unpack_scenario(packed, r)
        o = oracle.rollout(s["knot_off"], s["knots"], s["bbox"], s["etype"], s["kind"], s["ego"], s["t0"], s["length"], 1 / 30,
                           ctrl=s["ctrl"], max_steps=steps, force_steps=True)
        w = oracle.rss_rollout(o, s["bbox"], s["ego"])
        assert np.array_equal(codes[:, r, : len(s["bbox"])], w["code"]), r
        assert slong[r] == w["safe_longitudinal"] and slat[r] == w["safe_lateral"], r
        unsafe += not (w["safe_longitudinal"] and w["safe_lateral"])
    assert unsafe > 3


def test_rss_inside_rollout_equals_tick_by_tick(sga):
    """sg_set_rss: sg_rollout / sg_step run the callback themselves after the reset and after every step; flags, latest
    records and safe distances equal driving sg_rss_update by hand, also when scenarios finish at different times and when
    the call is repeated (a scenario that did not step is left alone)."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    R, E, steps = 40, 14, 90
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=L.KIND_AGENT_PID, extent=16.0)
    packed.length = packed.length * np.linspace(0.4, 1.0, R)  # ragged ends
    a = sga.RolloutEngine(R, E)
    a.upload(packed)
    a.rss_update(reset=True)
    for _ in range(steps + 5):
        a.lib.sg_rollout_async(a.h, 1, 0)
        a.rss_update()
    a.rss_update()  # nothing stepped: no change
    b = sga.RolloutEngine(R, E)
    b.set_rss(True)
    b.upload(packed)
    b.rollout(steps + 5)
    ra, rb = a.rss(), b.rss()
    assert (a.state()["n_steps"] == b.state()["n_steps"]).all() and len(set(a.state()["n_steps"])) > 5
    for x, y in zip(ra, rb):
        assert np.array_equal(x, y, equal_nan=True)
    assert not ra[0].all() or not ra[1].all()
    # a new batch on the same handle starts from empty histories
    packed2 = synthetic.make_batch(R, E, n_steps=steps, ego_kind=L.KIND_AGENT_PID, extent=16.0, seed=5)
    b.upload(packed2)
    b.rollout(steps)
    c = sga.RolloutEngine(R, E)
    c.set_rss(True)
    c.upload(packed2)
    c.rollout(steps)
    for x, y in zip(b.rss(), c.rss()):
        assert np.array_equal(x, y, equal_nan=True)
    a.close()
    b.close()
    c.close()


@pytest.mark.parametrize("R,E,ego", [(30, 3, "pid"), (12, 40, "replay"), (6, 100, "pid"), (4, 200, "replay"), (20, 64, "vehicle"),
                                     (3, 300, "pid"), (2, 512, "replay"), (2, 700, "pid"), (2, 600, "vehicle")])
def test_rss_fused_rollout_matches_oracle(sga, oracle, R, E, ego):
    """rollout_kernel_rss over the tile widths (4 ... 64 lanes) and the two- / four-wavefront scenarios, replay, PID and
    external-action egos: records of the latest update, safe distances and metric flags equal the oracle's callback run
    over the oracle's rollout; a second rollout() (reset included) gives the same again.  Scenarios of more than 512
    entities run the callback as a launch of its own after every step of the multi-kernel step: the same records."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic
    from scenario_gym_amd.packing import unpack_scenario

    steps = 70
    kind = dict(replay=L.KIND_AGENT_REPLAY, pid=L.KIND_AGENT_PID, vehicle=L.KIND_AGENT_VEHICLE)[ego]
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=kind, extent=20.0 if E < 100 else (45.0 if E <= 256 else 80.0), vanish_frac=0.3)
    packed.length = packed.length * np.linspace(0.5, 1.0, R)
    acts = synthetic.make_actions(steps, R) if ego == "vehicle" else None
    eng = sga.RolloutEngine(R, E)
    eng.set_rss(True)
    eng.upload(packed)
    for rep in range(2):
        if ego == "vehicle":
            eng.reset()
            eng.step(steps, acts)
        else:
            eng.rollout(steps)
        slong, slat, codes, safe = eng.rss()
        n_steps = eng.state()["n_steps"]
        for r in range(R):
            s = unpack_scenario(packed, r)
            kw = dict(actions=acts[:, r], force_steps=True) if ego == "vehicle" else {}
            o = oracle.rollout(s["knot_off"], s["knots"], s["bbox"], s["etype"], s["kind"], s["ego"], s["t0"], s["length"], 1 / 30,
                               ctrl=s["ctrl"], max_steps=steps, **kw)
            w = oracle.rss_rollout(o, s["bbox"], s["ego"])
            n = len(s["bbox"])
            assert n_steps[r] == o["n_steps"]
            assert np.array_equal(codes[r, :n], w["code"][-1]), (rep, r, codes[r, :n], w["code"][-1])
            assert np.array_equal(safe[r, :n], w["safe"][-1], equal_nan=True), (rep, r)
            assert slong[r] == w["safe_longitudinal"] and slat[r] == w["safe_lateral"], (rep, r)
    eng.close()


@pytest.mark.parametrize("scene,E", [("crowd", 12), ("crowd", 40), ("crowd", 150), ("roads", 6), ("roads", 40), ("roads", 100),
                                     ("crowd", 300), ("roads", 300)])
def test_rss_inside_pedestrian_and_off_road_rollouts(sga, scene, E):
    """sg_set_rss on batches with pedestrian agents (a PID car and a replayed car driving through a social-force crowd) and on
    batches with the ego_off_road terminal condition: the callback runs inside those rollout variants too
    (rollout_kernel_rss_ped / _road) and leaves what one step per launch + sg_rss_update by hand leaves.  (257..512 entities:
    no fused variant carries those combinations -- the library itself goes step by step with the callback behind every step.)"""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    rng = np.random.default_rng(31)
    R, steps, dt = 10, 80, 1 / 30
    nets = net_of = None
    if scene == "crowd":
        side = {12: 8.0, 40: 12.0, 150: 22.0, 300: 30.0}[E]
        packed = synthetic.make_crowd(R, E, n_steps=steps, side=side)
        T = steps * dt
        for r in range(R):
            for slot, (kind, y, v) in enumerate([(L.KIND_AGENT_PID, -1.0, 4.0), (L.KIND_REPLAY, 2.5, -3.0)]):
                i = r * E + slot
                a = int(packed.knot_off[i])
                x0 = -np.sign(v) * side / 2
                packed.knots[a] = [0.0, x0, y, 0.0, 0.0 if v > 0 else np.pi, 0.0, 0.0]
                packed.knots[a + 1] = [T, x0 + v * T, y, 0.0, 0.0 if v > 0 else np.pi, 0.0, 0.0]
                packed.kind[i], packed.etype[i] = kind, 0
                packed.bbox[i] = synthetic.CAR1_BBOX
                packed.ctrl[i] = sga.engine.DEFAULT_CTRL
        kw = {}
    else:
        packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=L.KIND_AGENT_PID, extent=25.0)
        nets = []
        for n in range(3):
            rings = []
            for q in range(6):
                c = rng.uniform(-30, 30, 2)
                ang = np.sort(rng.uniform(0, 2 * np.pi, 12))
                rings.append(c + (rng.uniform(25, 50) * rng.uniform(0.7, 1.0, 12))[:, None] * np.stack([np.cos(ang), np.sin(ang)], 1))
            vert_off = np.concatenate([[0], np.cumsum([len(r) for r in rings])])
            nets.append(dict(ring_off=np.arange(len(rings) + 1), vert_off=vert_off, verts=np.concatenate(rings), layers=np.ones(len(rings), int)))
        net_of = rng.integers(0, len(nets), R)
        kw = dict(terminal_conditions=["max_length", "ego_off_road"])
    a = sga.RolloutEngine(R, E, **kw)
    a.upload(packed)
    b = sga.RolloutEngine(R, E, **kw)
    b.set_rss(True)
    b.upload(packed)
    if nets:
        a.set_road_networks(nets, net_of)
        b.set_road_networks(nets, net_of)
    a.lib.sg_rollout_async(a.h, 0, 1)
    a.rss_update(reset=True)
    for _ in range(steps):
        a.lib.sg_rollout_async(a.h, 1, 0)
        a.rss_update()
    b.rollout(steps)
    if E <= 256:
        assert b.last_launch_stats()[0] == 1  # one launch, not one per step
    assert np.array_equal(a.state()["n_steps"], b.state()["n_steps"])
    if scene == "roads":
        assert len(set(a.state()["n_steps"])) > (2 if E <= 256 else 1)  # egos did leave the road at different times
    ra, rb = a.rss(), b.rss()
    for x, y in zip(ra, rb):
        assert np.array_equal(x, y, equal_nan=True)
    assert (ra[2] > 0).any()
    a.close()
    b.close()


@pytest.mark.parametrize("R,E,extent", [(16, 12, 10.0), (6, 640, 75.0)])
def test_masked_reset_restarts_the_rss_histories(sga, R, E, extent):
    """sg_reset_scenarios with the RSS callback on: the flagged scenarios' histories (the "unsafe" entries behind the metric
    flags, `last`) start anew with the reset-time update, the others carry on (also on scenarios of more than 512 entities)."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    packed = synthetic.make_batch(R, E, n_steps=60, ego_kind=L.KIND_AGENT_PID, extent=extent)
    mask = (np.arange(R) % 3 == 0).astype(np.uint8)
    runs = {}
    for name, plan in (("restarted", (30, "reset", 20)), ("fresh20", (20,)), ("straight50", (50,))):
        eng = sga.RolloutEngine(R, E)
        eng.set_rss(True)
        eng.upload(packed)
        for op in plan:
            if op == "reset":
                eng.reset_scenarios(mask)
            else:
                eng.step(op)
        runs[name] = eng.rss()
        eng.close()
    m = mask.astype(bool)
    assert not runs["straight50"][0][m].all() or not runs["straight50"][1][m].all()  # (some restarted scenario had an unsafe entry)
    for k in range(4):
        assert np.array_equal(runs["restarted"][k][m], runs["fresh20"][k][m], equal_nan=True), k
        assert np.array_equal(runs["restarted"][k][~m], runs["straight50"][k][~m], equal_nan=True), k


@pytest.mark.parametrize("R,E,extent", [(24, 20, 14.0), (3, 560, 70.0)])
def test_rss_callback_inside_the_graph_tick(sga, R, E, extent):
    """sg_tick (one captured launch per RL tick) with sg_set_rss: the captured step is the RSS variant + rss_lines_kernel (the
    multi-kernel step + rss_kernel on scenarios of more than 512 entities) -- the records after every tick equal those of
    sg_step; switching the callback on or off re-captures the graph."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    steps = 40
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=L.KIND_AGENT_VEHICLE, extent=extent)
    acts = synthetic.make_actions(steps, R)
    a = sga.RolloutEngine(R, E)
    b = sga.RolloutEngine(R, E)
    for eng in (a, b):
        eng.set_rss(True)
        eng.upload(packed)
    for k in range(steps):
        a.tick(acts[k], [0], nw=4, nh=4)
        b.step(1, acts[k:k + 1])
        if k % 13 == 0 or k == steps - 1:
            for x, y in zip(a.rss(), b.rss()):
                assert np.array_equal(x, y, equal_nan=True), k
    assert (a.rss()[2] > 0).any()
    a.set_rss(False)  # the next tick runs the plain step again
    before = [x.copy() for x in a.rss()]
    a.tick(acts[0], [0], nw=4, nh=4)
    for x, y in zip(a.rss(), before):
        assert np.array_equal(x, y, equal_nan=True)
    a.close()
    b.close()


@pytest.mark.parametrize("E,ego", [(64, "pid"), (9, "replay"), (130, "pid")])
def test_rss_line_test_queues_across_launches(sga, monkeypatch, E, ego):
    """The line tests of rollout_kernel_rss are queued per wavefront and evaluated after each launch (rss_lines_kernel) / after
    each work item of the persistent launch (rss_lines_block); the queues hold a fixed number of steps.  One piece, pieces
    of a few steps each (queues of 1 MiB), and a rollout resumed
    in pieces by the caller: the same records, states and flags -- the `last` entry and a pending "unsafe" class carry over
    from launch to launch."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    R, steps = 96, 80
    kind = dict(replay=L.KIND_AGENT_REPLAY, pid=L.KIND_AGENT_PID)[ego]
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=kind, extent=14.0 if E < 100 else 30.0, vanish_frac=0.2)
    packed.length = packed.length * np.linspace(0.5, 1.0, R)
    out = []
    for mode in ("one", "chunked", "resumed"):
        if mode == "chunked":
            monkeypatch.setenv("SG_RSSQ_MB", "1")
        else:
            monkeypatch.delenv("SG_RSSQ_MB", raising=False)
        eng = sga.RolloutEngine(R, E)
        eng.set_rss(True)
        eng.upload(packed)
        if mode == "resumed":
            eng.rollout(7)
            for n in (1, 20, 3, steps):
                eng.rollout_async(n, do_reset=False)
            eng.synchronize()
            launches = 0
        else:
            eng.rollout(steps)
            info = eng.schedule_info()  # pieces of the call: chunks of the persistent launch (PID egos), else launches
            launches = info["chunks"] if info["schedule"] == SCHED_QUEUE else eng.last_launch_stats()[0]
        out.append((eng.rss(), eng.state()["n_steps"].copy(), launches))
        eng.close()
    assert out[1][2] > out[0][2] >= 1  # the small queues did split the call
    for k in (1, 2):
        assert np.array_equal(out[0][1], out[k][1])
        for x, y in zip(out[0][0], out[k][0]):
            assert np.array_equal(x, y, equal_nan=True), k
    assert (out[0][0][2] >= 4).any() and (out[0][0][2] == 1).any() and (out[0][0][2] == 2).any()  # unsafe, lateral, longitudinal


@pytest.mark.parametrize("E,n_pid", [(5, 1), (64, 7), (20, 20), (200, 9)])
def test_rss_rollout_with_table_lanes_equals_in_kernel_controllers(sga, monkeypatch, E, n_pid):
    """rollout_kernel_rss_tab (controlled lanes read the poses control_kernel wrote; the default for a long rollout) against
    rollout_kernel_rss (the PID controllers inside the rollout kernel; SG_RSS_TAB=0): entity states, controller states, the
    ego's metric recurrences, step counts, events and every RSS record are the same bits, with one and with many PID lanes per
    scenario, some of them not the ego."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    R, steps = 70, 90
    packed = synthetic.make_batch(R, E, n_steps=steps, ego_kind=L.KIND_AGENT_PID, extent=14.0 if E < 100 else 40.0, vanish_frac=0.2)
    packed.length = packed.length * np.linspace(0.4, 1.0, R)
    kind = packed.kind.reshape(R, E)
    for r in range(R):
        lanes = [e for e in range(E) if kind[r, e] == L.KIND_REPLAY][:n_pid - 1]
        kind[r, lanes] = L.KIND_AGENT_PID
    out = []
    for mode in ("1", "0"):
        monkeypatch.setenv("SG_RSS_TAB", mode)
        eng = sga.RolloutEngine(R, E)
        eng.set_rss(True)
        eng.upload(packed)
        eng.rollout(steps)
        m, ev = eng.metrics()
        out.append((eng.rss(), eng.state(), m, ev, eng.collision_points()))
        eng.close()
    a, b = out
    for x, y in zip(a[0], b[0]):
        assert np.array_equal(x, y, equal_nan=True)
    for k in a[1]:
        assert np.array_equal(a[1][k], b[1][k], equal_nan=True), k
    assert a[2].tobytes() == b[2].tobytes() and a[3].tobytes() == b[3].tobytes()
    assert np.array_equal(a[4], b[4], equal_nan=True)
    assert (a[2]["ego_distance_travelled"] > 0).all()


# --------------------------------------------------------------------------- crowds with riders
_ALL_RIDERS = [("pid", 0, "car"), ("vehicle", 0, "car"), ("replay", 0, "car-leaves"), ("replay", 2, "static"),
               ("agent_replay", 0, "late"), ("replay", 1, "ped"), ("replay", 1, "ped")]


def _add_riders(packed, rng, side, T, spec):
    """Turn the first len(spec) slots of every scenario of a make_crowd batch into riders: (kind, catalog type code, what) --
    cars that cross the square (one leaves early, one appears late), a static obstacle (ONE knot), recorded pedestrians."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic
    from scenario_gym_amd.engine import DEFAULT_CTRL

    kinds = dict(pid=L.KIND_AGENT_PID, vehicle=L.KIND_AGENT_VEHICLE, replay=L.KIND_REPLAY, agent_replay=L.KIND_AGENT_REPLAY)
    R, E = packed.n_scenarios, packed.n_entities
    knots = packed.knots.reshape(R * E, 2, 7).copy()
    keep = np.ones(R * E, bool)
    static = np.zeros(R * E, bool)
    for r in range(R):
        for slot, (kind, etype, what) in enumerate(spec):
            i = r * E + slot
            keep[i] = False
            y = rng.uniform(-side / 3, side / 3)
            v = rng.choice([-1.0, 1.0]) * rng.uniform(1.5, 4.0)
            x0 = -np.sign(v) * side / 2
            h = 0.0 if v > 0 else np.pi
            t_a, t_b = 0.0, T
            if what == "car-leaves":
                t_b = 0.6 * T
            if what == "late":
                t_a = 0.25 * T
            knots[i, 0] = [t_a, x0, y, 0.0, h, 0.0, 0.0]
            knots[i, 1] = [t_b, x0 + v * (t_b - t_a), y, 0.0, h, 0.0, 0.0]
            static[i] = what == "static"
            packed.kind[i], packed.etype[i] = kinds[kind], etype
            if etype != 1:
                packed.bbox[i] = synthetic.CAR1_BBOX if etype == 0 else [1.0, 1.5, 0.2, 0.0]
            packed.ctrl[i] = DEFAULT_CTRL
    # the static obstacle has ONE knot: rebuild the ragged knot array; riders have no route
    rows = [knots[i, :1] if static[i] else knots[i] for i in range(R * E)]
    packed.knots = np.concatenate(rows)
    packed.knot_off = np.concatenate([[0], np.cumsum([len(x) for x in rows])]).astype(np.int64)
    packed.routes = packed.routes.reshape(R * E, 2, 2)[keep].reshape(-1, 2)
    packed.route_off = np.concatenate([[0], np.cumsum(np.where(keep, 2, 0))]).astype(np.int64)
    return packed.validate()


def _random_crowds(n, seed=77):
    rng = np.random.default_rng(seed)
    rrng = np.random.default_rng([seed, 9])  # (its own stream: earlier configurations keep their draws)
    brng = np.random.default_rng([seed, 10])
    wrng = np.random.default_rng([seed, 11])
    out = []
    for k in range(n):
        E = int([20, 64, 100, 256, 40, 130][k % 6])
        out.append(dict(E=E, R=int(rng.integers(2, 7 if E <= 64 else 4)), steps=int(rng.integers(30, 90)),
                        side=float(rng.choice([6.0, 12.0, 25.0])), roads=bool(rng.integers(0, 2)), seed=int(rng.integers(1, 1 << 30)),
                        dt=float(rng.choice([1 / 30, 0.1])), noise=str(rng.choice(["off", "off", "device", "stream"])),
                        radii=bool(rng.integers(0, 2)), late=bool(rng.integers(0, 3) == 0),
                        riders=int(rrng.integers(0, 8)) if rrng.integers(0, 2) else 0,  # how many of _ALL_RIDERS ride along
                        walk=bool(brng.integers(0, 5) == 0)))                            # RandomWalk instead of SocialForce
        # a third of the configurations without riders: routes of 2 ... 6 waypoints side by side (the goal update from registers
        # up to four, from device memory beyond)
        out[-1]["routes"] = bool(wrng.integers(0, 3) == 0) and out[-1]["riders"] == 0
        if brng.integers(0, 15) == 0:  # one in fifteen: a crowd of more than 512 (the multi-kernel step; road networks, both noise
            c = out[-1]                # modes and riders of every kind stay)
            c.update(E=int(brng.choice([520, 640])), R=int(brng.integers(1, 3)), steps=min(c["steps"], 40),
                     side=float(brng.choice([25.0, 40.0])))
    return out


# (two fixed ones beside the random draw: crowds of more than 512 pedestrians ON road networks -- the boundary terms at that width)
_WIDE_ROAD_CROWDS = [dict(E=600, R=2, steps=30, side=25.0, roads=True, seed=424243, dt=1 / 30, noise="off", radii=True, late=False, riders=0, walk=False),
                     dict(E=530, R=3, steps=25, side=40.0, roads=True, seed=424244, dt=0.1, noise="device", radii=False, late=True, riders=3, walk=False),
                     dict(E=700, R=2, steps=28, side=25.0, roads=False, seed=424245, dt=1 / 30, noise="stream", radii=True, late=True, riders=2, walk=False),
                     dict(E=520, R=2, steps=20, side=40.0, roads=True, seed=424246, dt=0.1, noise="stream", radii=False, late=False, riders=0, walk=True)]


@pytest.mark.parametrize("cfg", _random_crowds(int(os.environ.get("SG_FUZZ_CROWDS", "8")), int(os.environ.get("SG_FUZZ_SEED", "77"))) + _WIDE_ROAD_CROWDS,
                         ids=lambda c: f"E{c['E']}-s{c['side']:.0f}-{'roads' if c['roads'] else 'free'}-{c['noise']}{'-walk' if c['walk'] else ''}")
def test_randomized_crowds_match_oracle(sga, oracle, cfg):
    """Random social-force crowds (tile widths up to four wavefronts, sparse to packed), half of them on a road network with
    random convex buildings and pavements among the pedestrians: poses of every step, forces, collision rows and events
    bit-identical to the oracle (crowd kernel and the general pedestrian variant, balanced pair loops, boundary terms,
    density-adaptive broad phase); a quarter each with the counter-based noise generator and with a stream of variates,
    half with per-pedestrian sensor radii, a third with pedestrians that join the scene late (the spawn rule), a fifth with
    RandomWalk as the behaviour model (on road networks too: it ignores them, as the reference's does)."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic
    from scenario_gym_amd.packing import unpack_scenario

    rng = np.random.default_rng(cfg["seed"])
    R, E, steps, dt = cfg["R"], cfg["E"], cfg["steps"], cfg["dt"]
    packed = synthetic.make_crowd(R, E, n_steps=steps, timestep=dt, side=cfg["side"], seed=cfg["seed"] % 1000)
    if cfg["radii"]:
        packed.ctrl[:, L.C_PED_RADIUS] = rng.uniform(0.8, 3.5, R * E)
    if cfg["late"]:  # every fifth pedestrian's trajectory starts later: it is not in the scene at the reset and spawns
        kn = packed.knots.reshape(R * E, 2, 7)
        kn[::5, 0, 0] = rng.uniform(0.2, 1.0, len(kn[::5])) * steps * dt * 0.5
    if cfg.get("routes"):
        wr = np.random.default_rng([cfg["seed"], 17])
        two = packed.routes.reshape(R * E, 2, 2)
        routes, off = [], [0]
        for i in range(R * E):
            n = int(wr.integers(2, 7))
            frac = np.sort(wr.uniform(0.0, 1.0, n - 2))
            mid = two[i, 0] + frac[:, None] * (two[i, 1] - two[i, 0]) + wr.normal(0, 0.08 * cfg["side"], (n - 2, 2))
            routes.append(np.concatenate([two[i, :1], mid, two[i, 1:]]))
            off.append(off[-1] + n)
        packed.routes, packed.route_off = np.concatenate(routes), np.array(off, np.int64)
        packed.validate()
    if cfg["riders"]:  # cars / obstacles / recorded pedestrians among the walkers (64-lane tiles without roads: the riders' path)
        _add_riders(packed, np.random.default_rng([cfg["seed"], 13]), cfg["side"], steps * dt, _ALL_RIDERS[: cfg["riders"]])
    noise_kw, noise_o = {}, [None] * R
    if cfg["noise"] == "device":
        noise_kw = dict(std_lon=0.1, std_lat=0.05, noise="device", noise_seed=cfg["seed"])
        noise_o = [dict(mode="device", std_lon=0.1, std_lat=0.05, seed=cfg["seed"], scenario_index=r) for r in range(R)]
    elif cfg["noise"] == "stream":
        normals = np.random.RandomState(cfg["seed"] % (1 << 31)).standard_normal((R, 2 * E * (steps + 1)))
        noise_kw = dict(std_lon=0.05, std_lat=0.1, noise="stream", normals=normals)
        noise_o = [dict(mode="stream", std_lon=0.05, std_lat=0.1, normals=normals[r]) for r in range(R)]
    nets, net_of = [], np.full(R, -1, np.int32)
    if cfg["roads"]:
        for n in range(2):
            rings, layers = [], []
            for q in range(int(rng.integers(2, 7))):
                c = rng.uniform(-cfg["side"], cfg["side"], 2)
                m = int(rng.integers(3, 9))
                ang = np.sort(rng.uniform(0, 2 * np.pi, m))
                rad = rng.uniform(0.8, 0.35 * cfg["side"] + 1.0)
                rings.append(c + rad * np.stack([np.cos(ang), np.sin(ang)], 1))
                layers.append(int(rng.choice([16 | 128, 16 | 128, 16 | 32])))  # building (walkable + impenetrable) or pavement
            nets.append(dict(ring_off=np.arange(len(rings) + 1), vert_off=np.concatenate([[0], np.cumsum([len(r) for r in rings])]),
                             verts=np.concatenate(rings), layers=np.array(layers)))
        net_of = rng.integers(-1, 2, R).astype(np.int32)
    behaviour, sf = "social_force", None
    if cfg["walk"]:  # a fifth of the configurations: every pedestrian follows RandomWalk (pedestrian/random_walk.py), with a bias
        behaviour, sf = "random_walk", oracle.social_force_params(bias_lon=0.1, bias_lat=-0.03)
        noise_kw = dict(noise_kw, behaviour="random_walk", bias_lon=0.1, bias_lat=-0.03)
    eng = sga.RolloutEngine(R, E, timestep=dt, record_capacity=steps + 1, event_capacity=512, social_force=noise_kw or None)
    eng.upload(packed)
    if nets:
        eng.set_road_networks(nets, net_of)
    eng.rollout(steps)
    st = eng.state()
    rows, events = eng.metrics()
    t, poses = eng.record(steps + 1)
    eng.close()
    for r in range(R):
        s = unpack_scenario(packed, r)
        road = nets[net_of[r]] if (nets and net_of[r] >= 0) else None
        o = oracle.rollout(s["knot_off"], s["knots"], s["bbox"], s["etype"], s["kind"], s["ego"], s["t0"], s["length"], dt,
                           ctrl=s["ctrl"], route_off=s["route_off"], routes=s["routes"], max_steps=steps, event_cap=512, road=road,
                           noise=noise_o[r], actions=np.zeros((steps, 2)) if (s["kind"] == L.KIND_AGENT_VEHICLE).any() else None,
                           sf=sf, behaviour=behaviour)
        n = o["n_steps"]
        assert rows["n_steps"][r] == n, r
        assert bits_equal(poses[: n + 1, r], o["poses"]), (r, "poses")
        ped = s["kind"] == L.KIND_AGENT_PEDESTRIAN   # (the oracle's extra columns are per kind: riders keep controller state there)
        assert bits_equal(st["force"][r][ped], o["extra"][-1, ped, 2:]) and bits_equal(st["dists"][r], o["dists"][-1]), r
        assert bits_equal(st["ctrl_state"][r][~ped], o["extra"][-1, ~ped]), r
        assert np.array_equal(_dense_words(st["coll"][r], E), oracle.coll_to_dense(o["coll"], E)[-1]), r
        ev = events[events["scenario"] == r]
        m = min(len(ev), 512)
        assert rows["n_collisions"][r] == o["n_events"] and np.array_equal(ev["t"][:m], o["ev_t"][:m]), r


@pytest.mark.parametrize("cfg", _random_configs(int(os.environ.get("SG_FUZZ_RSS", "12")), int(os.environ.get("SG_FUZZ_SEED", "2024")) + 1),
                         ids=lambda c: f"E{c['E']}-{c['ego']}-{'p' if c['persist'] else 'n'}-{len(c['terminal'])}{c['terminal'][-1][0]}")
def test_randomized_rss_matches_oracle(sga, oracle, cfg):
    """The RSS callback inside the rollout kernel over the configuration space of the randomized sweep (tile widths, persist,
    terminal conditions that end scenarios early, ego kinds, time steps): records of the latest update, safe distances and
    metric flags equal the oracle's callback run along the oracle's rollout."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic
    from scenario_gym_amd.engine import TERMINAL_BITS

    kind = dict(replay=L.KIND_AGENT_REPLAY, pid=L.KIND_AGENT_PID, vehicle=L.KIND_AGENT_VEHICLE)[cfg["ego"]]
    R, E, steps, dt = cfg["R"], cfg["E"], cfg["steps"], cfg["dt"]
    packed = synthetic.make_batch(R, E, n_steps=steps, timestep=dt, n_knots=cfg["knots"], ego_kind=kind,
                                  static_frac=cfg["static"], vanish_frac=cfg["vanish"], extent=cfg["extent"], seed=cfg["seed"])
    force = cfg["ego"] == "vehicle"
    acts = synthetic.make_actions(steps, R, seed=cfg["seed"]) if force else None
    eng = sga.RolloutEngine(R, E, timestep=dt, persist=cfg["persist"], terminal_conditions=cfg["terminal"])
    eng.set_rss(True)
    eng.upload(packed)
    if force:
        eng.step(steps, acts)
    else:
        eng.rollout(steps)
    slong, slat, codes, safe = eng.rss()
    n_steps = eng.state()["n_steps"]
    eng.close()
    mask = sum(TERMINAL_BITS[c] for c in cfg["terminal"])
    for r in range(R):
        kw = dict(actions=acts[:, r], force_steps=True) if force else {}
        o = _oracle_one(oracle, packed, r, dt, steps, persist=cfg["persist"], terminal_mask=mask, event_cap=64, **kw)
        from scenario_gym_amd.packing import unpack_scenario
        s = unpack_scenario(packed, r)
        w = oracle.rss_rollout(o, s["bbox"], s["ego"])
        n = len(s["bbox"])
        assert n_steps[r] == o["n_steps"], r
        assert np.array_equal(codes[r, :n], w["code"][-1]), (r, codes[r, :n], w["code"][-1])
        assert np.array_equal(safe[r, :n], w["safe"][-1], equal_nan=True), r
        assert slong[r] == w["safe_longitudinal"] and slat[r] == w["safe_lateral"], r


# ---------------------------------------------------------------- time-sliced replay (sg_set_slicing)
def _final_results(eng, steps):
    eng.rollout(steps)
    st = eng.state()
    rows, events = eng.metrics()
    return st, rows, events


@pytest.mark.parametrize("R,E,steps,terminal,persist,dt", [
    (256, 16, 400, ["max_length"], False, 1 / 30),                   # BASELINE config 2 shape
    (96, 5, 333, ["max_length", "collision"], False, 1 / 30),        # scenarios end at different steps (first collision)
    (70, 30, 250, ["max_length", "ego_collision"], True, 0.1),       # persist, tile of 32 lanes, ego collisions end some
    (40, 64, 200, ["max_length"], False, 1 / 30),                    # full 64-lane tiles
    (33, 3, 777, ["max_length"], False, 0.02),                       # tile of 4 lanes, slices of unequal length
])
def test_sliced_replay_equals_stepwise_and_oracle(sga, oracle, R, E, steps, terminal, persist, dt):
    """sg_rollout of a replay-only batch with the time axis cut into slices (clock, slices side by side, last step
    materialised, ordered sums): final state, metrics, events and the step each scenario stopped at are bit-identical to the
    step-by-step kernel and to the oracle.  Event lists longer than the capacity (2 here) keep their first entries."""
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    packed = synthetic.make_batch(R, E, n_steps=steps, timestep=dt, ego_kind=L.KIND_AGENT_REPLAY, static_frac=0.15,
                                  vanish_frac=0.2, extent=25.0 if E > 8 else 12.0)
    out = {}
    for mode in (False, "always"):
        eng = sga.RolloutEngine(R, E, timestep=dt, persist=persist, terminal_conditions=terminal, event_capacity=2)
        eng.set_slicing(mode)
        eng.upload(packed)
        out[mode] = _final_results(eng, steps)
        if mode == "always":  # again on the same handle (its slice arrays are reused), then stepwise continues from the state
            again = _final_results(eng, steps)
            for k in ("poses", "vels", "dists", "coll", "t", "prev_t", "n_steps", "done", "present"):
                assert bits_equal(again[0][k], out[mode][0][k]) if again[0][k].dtype.kind == "f" else np.array_equal(again[0][k], out[mode][0][k]), k
            assert np.array_equal(again[1], out[mode][1]) and np.array_equal(again[2], out[mode][2])
        eng.close()
    (sa, ra, ea), (sb, rb, eb) = out[False], out["always"]
    for k in ("poses", "vels", "dists", "t", "prev_t"):
        assert bits_equal(sa[k], sb[k]), k
    for k in ("coll", "n_steps", "done", "present"):
        assert np.array_equal(sa[k], sb[k]), k
    assert np.array_equal(ra, rb) and np.array_equal(ea, eb)
    assert len(ea) > 0 and ((ra["n_collisions"] > 2).any() or len(terminal) > 1 or E < 16)  # some lists overflow the capacity of 2
    if len(terminal) > 1:
        assert len(set(ra["n_steps"])) > 3  # scenarios do stop at different steps
    mask = sum(dict(max_length=1, collision=2, ego_collision=4)[c] for c in terminal)
    for r in range(0, R, 7):
        o = _oracle_batch(oracle, packed, dt, steps, [r], persist=persist, terminal_mask=mask)[r]
        assert rb["n_steps"][r] == o["n_steps"] and rb["final_t"][r] == o["final_t"] and bool(rb["done"][r]) == o["is_done"], r
        assert bits_equal(sb["poses"][r], o["poses"][-1]) and bits_equal(sb["vels"][r], o["vels"][-1]), r
        assert bits_equal(sb["dists"][r], o["dists"][-1]) and np.array_equal(sb["coll"][r], o["coll"][-1, :, 0]), r
        for k in ("ego_avg_speed", "ego_max_speed", "ego_distance_travelled"):
            assert rb[k][r] == o["metric_" + k] or (np.isnan(rb[k][r]) and np.isnan(o["metric_" + k])), (r, k)
        ev = eb[eb["scenario"] == r]
        assert rb["n_collisions"][r] == o["n_events"] and np.array_equal(ev["t"], o["ev_t"][:2]) and np.array_equal(ev["other"], o["ev_other"][:2]), r


@pytest.mark.parametrize("R,E,steps,terminal,dt,ego", [
    (64, 64, 700, ["max_length"], 1 / 30, "pid"),                       # config 4's shard shape in small: 64-lane tiles, PID egos
    (48, 30, 420, ["max_length", "ego_collision"], 1 / 30, "pid"),      # tiles of 32 lanes, scenarios stop at different steps
    (40, 16, 333, ["max_length", "collision"], 0.1, "vehicle"),         # tiles of 16 lanes, VehicleController egos (zero actions)
    (36, 64, 515, ["max_length"], 0.05, "late"),                        # a PID agent that is NOT the ego and spawns at step 1
])
def test_sliced_rollout_with_controlled_lanes(sga, oracle, R, E, steps, terminal, dt, ego):
    """sg_rollout time-sliced for batches WITH controlled lanes (BASELINE config 4's shards: 512 x 64 with PID egos): the
    controller pre-pass fills one table for the whole call, the slices replay it group by group.  Final state, controller
    state, metrics, events (with their classes and collision points) and the step each scenario stopped at are bit-identical
    to the step-by-step path and to the oracle."""
    import scenario_gym_amd._lib as L
    from oracle import check
    from scenario_gym_amd import synthetic

    kind = dict(pid=L.KIND_AGENT_PID, vehicle=L.KIND_AGENT_VEHICLE, late=L.KIND_AGENT_REPLAY)[ego]
    packed = synthetic.make_batch(R, E, n_steps=steps, timestep=dt, ego_kind=kind, static_frac=0.15, vanish_frac=0.25,
                                  extent=30.0 if E > 16 else 14.0)
    if ego == "late":  # the first entity of each scenario that starts after t0 becomes the (only) PID agent
        n_late = 0
        for r in range(R):
            for e in range(1, E):
                i = r * E + e
                a, b = packed.knot_off[i], packed.knot_off[i + 1]
                if b - a > 1 and packed.knots[a, 0] > packed.t0[r] + 3 * dt:
                    packed.kind[i] = L.KIND_AGENT_PID
                    n_late += 1
                    break
        assert n_late > R // 2
    mask = sum(dict(max_length=1, collision=2, ego_collision=4)[c] for c in terminal)
    out = {}
    for mode in (False, "always", True, "general"):
        eng = sga.RolloutEngine(R, E, timestep=dt, terminal_conditions=terminal, event_capacity=8)
        eng.set_slicing("always" if mode == "general" else mode)
        if mode == "general":  # the pre-pass of the sliced path without its straight-line fast block (read at upload)
            os.environ["SG_CTL_FAST"] = "0"
        try:
            eng.upload(packed)
        finally:
            os.environ.pop("SG_CTL_FAST", None)
        st, rows, events = _final_results(eng, steps)
        pts = eng.collision_points()
        n_launch = eng.last_launch_stats()[0]
        if mode:
            ver = check.verify_engine(eng, packed, dt, steps, K=R, event_cap=8, terminal_mask=mask)
            assert ver["equal"], ver["mismatches"]
            st2, rows2, events2 = _final_results(eng, steps)  # again on the same handle: its arrays are reused
            for name in rows.dtype.names:
                assert rows[name].tobytes() == rows2[name].tobytes(), (mode, "rows", name, np.flatnonzero(rows[name] != rows2[name])[:8])
            assert len(events) == len(events2), (mode, len(events), len(events2))
            for name in events.dtype.names:
                assert events[name].tobytes() == events2[name].tobytes(), (mode, "events", name, np.flatnonzero(events[name] != events2[name])[:8])
            assert rows.tobytes() == rows2.tobytes() and events.tobytes() == events2.tobytes()
            for name in ("poses", "vels", "dists", "ctrl_state", "present", "coll"):
                a, b = np.asarray(st[name]), np.asarray(st2[name])
                if a.tobytes() != b.tobytes():
                    bad = np.argwhere(a.view(np.uint64 if a.dtype.itemsize == 8 else a.dtype) != b.view(np.uint64 if b.dtype.itemsize == 8 else b.dtype))[:6]
                    raise AssertionError((mode, name, bad.tolist(), [(a[tuple(i)], b[tuple(i)]) for i in bad],
                                          rows["n_steps"][bad[:, 0]].tolist(), rows["done"][bad[:, 0]].tolist()))
        out[mode] = (st, rows, events, pts, n_launch)
        eng.close()
    assert out["always"][4] > 3 and out[True][4] >= 1   # several groups of short slices / the default plan
    for mode in ("always", True, "general"):
        (sa, ra, ea, pa, _), (sb, rb, eb, pb, _) = out[False], out[mode]
        for k in ("poses", "vels", "dists", "t", "prev_t", "ctrl_state"):
            assert bits_equal(sa[k], sb[k]), (mode, k)
        for k in ("coll", "n_steps", "done", "present"):
            assert np.array_equal(sa[k], sb[k]), (mode, k)
        assert ra.tobytes() == rb.tobytes() and ea.tobytes() == eb.tobytes(), mode
        assert np.array_equal(pa, pb, equal_nan=True), mode
    assert len(out[False][2]) > 0
    if len(terminal) > 1:
        assert len(set(out[False][1]["n_steps"])) > 3


def test_sliced_replay_on_the_reference_scenarios(sga, oracle):
    """The 23 OpenSCENARIO files of the reference's tests as one ragged batch (1 ... 9 entities, different lengths and start
    times, egos that appear late), sliced: clock, final poses / velocities / distances / collisions and the ego metrics are
    the real reference's (golden all_scenarios), bit for bit."""
    from scenario_gym_amd.packing import default_kinds, pack_arrays

    g = load_golden("all_scenarios")
    names = list(g["names"])
    scs = []
    for n in names:
        s = scenario_arrays(g, f"{n}/scenario")
        s["kind"] = default_kinds(len(s["bbox"]), s["ego"])
        scs.append(s)
    packed = pack_arrays(scs)
    n_max = max(len(g[f"{n}/t"]) for n in names) + 4
    eng = sga.RolloutEngine(packed.n_scenarios, packed.n_entities, timestep=1 / 30)
    eng.set_slicing("always")
    eng.upload(packed)
    st, rows, events = _final_results(eng, n_max)
    eng.close()
    for i, n in enumerate(names):
        ts = g[f"{n}/t"]
        assert st["t"][i] == ts[-1] and rows["n_steps"][i] == len(ts) - 1 and rows["done"][i], n
        final = g[f"{n}/final_poses"]
        E = len(final)
        assert bits_equal(st["poses"][i, :E], final) and bits_equal(st["vels"][i, :E], np.where(np.isnan(final[:, :1]), np.nan, g[f"{n}/final_vels"])), n
        assert bits_equal(st["dists"][i, :E], g[f"{n}/final_dists"]), n
        assert np.array_equal(_dense(st["coll"][i, :E], E), g[f"{n}/final_coll"]), n
        for key in ("ego_avg_speed", "ego_max_speed", "ego_distance_travelled"):
            assert rows[key][i] == float(g[f"{n}/metric_{key}"]), (n, key)


# --------------------------------------------------------------------------- the timed shapes, oracle-checked
def test_timed_shape_c3_full_horizon_32_scenarios(sga, oracle):
    """BASELINE config 3 exactly as bench.py times it -- 4096 scenarios x 64 entities x 10,000 steps, PID egos, max_length,
    event capacity 64 -- with 32 scenarios spread over the batch re-run through the oracle for the FULL horizon: step counts,
    clock, every entity's final pose / velocity / distance / collision row / controller state, the ego metric rows and the
    event lists, bit for bit.  (the timed configuration itself, not a narrower or shorter stand-in.)"""
    import scenario_gym_amd._lib as L
    from oracle import check
    from scenario_gym_amd import synthetic

    R, E, T = 4096, 64, 10000
    packed = synthetic.make_batch(R, E, n_steps=T, ego_kind=L.KIND_AGENT_PID)
    eng = sga.RolloutEngine(R, E, terminal_conditions=["max_length"], event_capacity=64)
    eng.set_slicing(False)
    eng.upload(packed)
    eng.rollout(T)
    ver = check.verify_engine(eng, packed, 1 / 30, T, K=32, event_cap=64, threads=16)
    rows, _ = eng.metrics()
    eng.close()
    assert ver["scenarios"] == 32 and ver["equal"], ver["mismatches"]
    assert (rows["n_steps"] >= T - 1).all() and rows["done"].all()


def test_timed_shape_c5_crowd_full_horizon_8_scenarios(sga, oracle):
    """BASELINE config 5 at its timed width AND length -- 1024 scenarios x 256 pedestrians x 10,000 steps, the crowd kernel,
    exactly what `bench.py --workload c5` times -- 8 scenarios spread over the batch against the oracle over the full horizon
    (through the densest phase, the dispersal and the thousands of steps in which most pedestrians have arrived and a few
    are stuck): poses, velocities, distances, collision rows (4 words per entity), social forces, ego metrics, events."""
    from oracle import check
    from scenario_gym_amd import synthetic

    R, E, T = 1024, 256, 10000
    packed = synthetic.make_crowd(R, E, n_steps=T)
    eng = sga.RolloutEngine(R, E, terminal_conditions=["max_length"], event_capacity=64)
    eng.upload(packed)
    eng.rollout(T)
    ver = check.verify_engine(eng, packed, 1 / 30, T, K=8, event_cap=64, ped=True, threads=16)
    rows, _ = eng.metrics()
    eng.close()
    assert ver["scenarios"] == 8 and ver["equal"], ver["mismatches"]
    assert (rows["n_steps"] >= T - 1).all() and rows["done"].all()


def test_crowd_with_a_car_matches_oracle(sga, oracle):
    """bench.py --workload c5mix in small: crowds with one PID car each (the general pedestrian variant, not the crowd kernel),
    64- and 256-lane tiles, every scenario against the oracle."""
    from oracle import check
    from scenario_gym_amd import synthetic

    for R, E, T, side in ((12, 40, 150, 10.0), (4, 200, 120, 26.0)):
        packed = synthetic.make_crowd_with_car(R, E, n_steps=T, side=side)
        eng = sga.RolloutEngine(R, E, terminal_conditions=["max_length"], event_capacity=32)
        eng.upload(packed)
        eng.rollout(T)
        ver = check.verify_engine(eng, packed, 1 / 30, T, K=R, event_cap=32, ped=True)
        rows, events = eng.metrics()
        eng.close()
        assert ver["equal"], ver["mismatches"]
        assert len(events) > 0   # the car does run into pedestrians


@pytest.mark.parametrize("E,side,persist,terminal", [
    (64, 10.0, False, ["max_length"]),
    (150, 20.0, True, ["max_length"]),
    (256, 26.0, False, ["max_length", "ego_collision"]),
    (40, 7.0, False, ["max_length", "collision"]),
])
def test_crowd_riders_equal_general_variant_and_oracle(sga, oracle, monkeypatch, E, side, persist, terminal):
    """Crowds with RIDERS -- a PID car (the ego), a car on external actions (zero here: sg_rollout), a replayed car that leaves
    before the end, a static obstacle, a replay AGENT that spawns late, and two RECORDED pedestrians (replay entities of
    type Pedestrian: social-force neighbours of the walking ones) -- run through rollout_kernel_crowd_riders with the riders'
    poses from control_kernel_riders.  Poses of every step, forces, controller state, collision rows, metrics, events and
    classes equal the general pedestrian variant (SG_CROWD_RIDERS=0) and the oracle, bit for bit; resumed in pieces too."""
    import scenario_gym_amd._lib as L
    from oracle import check
    from scenario_gym_amd import synthetic
    from scenario_gym_amd.engine import DEFAULT_CTRL, TERMINAL_BITS

    R, steps, dt = 6, 100, 1 / 30
    packed = synthetic.make_crowd(R, E, n_steps=steps, side=side)
    _add_riders(packed, np.random.default_rng(E), side, steps * dt, _ALL_RIDERS)
    mask = sum(TERMINAL_BITS[c] for c in terminal)
    out = {}
    for mode in ("riders", "general", "pieces"):
        monkeypatch.setenv("SG_CROWD_RIDERS", "0" if mode == "general" else "1")
        eng = sga.RolloutEngine(R, E, persist=persist, terminal_conditions=terminal, record_capacity=steps + 1, event_capacity=64)
        eng.upload(packed)
        if mode == "pieces":   # reset + 3 resumed calls (the table restarts with every call)
            eng.rollout(37)
            eng.rollout_async(41, do_reset=False)
            eng.rollout_async(steps - 78, do_reset=False)
            eng.synchronize()
        else:
            eng.rollout(steps)
        st, (rows_, ev), (tt, poses), pts = eng.state(), eng.metrics(), eng.record(steps + 1), eng.collision_points()
        if mode == "riders":
            ver = check.verify_engine(eng, packed, dt, steps, K=R, event_cap=64, ped=True, persist=persist, terminal_mask=mask)
            assert ver["equal"], ver["mismatches"]
        out[mode] = (st, rows_, ev, poses, pts)
        eng.close()
    for mode in ("general", "pieces"):
        (sa, ra, ea, pa, qa), (sb, rb, eb, pb, qb) = out["riders"], out[mode]
        assert bits_equal(pa, pb), mode
        for k in ("poses", "vels", "dists", "force", "ctrl_state", "t"):
            assert bits_equal(sa[k], sb[k]), (mode, k)
        assert np.array_equal(sa["coll"], sb["coll"]) and np.array_equal(sa["present"], sb["present"]), mode
        assert ra.tobytes() == rb.tobytes() and len(ea) == len(eb), mode
        # (in tiles of several wavefronts the general variant cannot keep the pose of a hazard that is itself a controlled
        # agent -- its events stay unclassified, type -2, no collision point; the riders' table has it)
        open_ = (eb["type"] == -2) if mode == "general" else np.zeros(len(eb), bool)
        assert E > 64 or not open_.any()
        for f in ("t", "scenario", "other"):
            assert np.array_equal(ea[f], eb[f]), (mode, f)
        assert np.array_equal(ea["type"][~open_], eb["type"][~open_]) and np.array_equal(qa[~open_], qb[~open_], equal_nan=True), mode
    # gym.step() with external actions for the VehicleController rider (integrations/openaigym.py:197-204): same bits both ways
    acts = synthetic.make_actions(steps, R, seed=E)
    stepped = {}
    for mode in ("riders", "general"):
        monkeypatch.setenv("SG_CROWD_RIDERS", "0" if mode == "general" else "1")
        eng = sga.RolloutEngine(R, E, persist=persist, terminal_conditions=terminal, record_capacity=steps + 1, event_capacity=64)
        eng.upload(packed)
        eng.step(30, acts[:30])
        eng.step(1, acts[30:31])
        eng.step(steps - 31, acts[31:])
        stepped[mode] = (eng.state(), eng.metrics()[0], eng.record(steps + 1)[1])
        eng.close()
    assert bits_equal(stepped["riders"][2], stepped["general"][2]) and stepped["riders"][1].tobytes() == stepped["general"][1].tobytes()
    for k in ("poses", "vels", "dists", "force", "ctrl_state"):
        assert bits_equal(stepped["riders"][0][k], stepped["general"][0][k]), k
    assert not bits_equal(stepped["riders"][2], out["riders"][3])   # (the actions did steer the second car)
    st, rows_, ev, poses, _ = out["riders"]
    assert len(ev) > 0
    for r in range(R):  # every recorded step against the oracle as well
        o = _oracle_one(oracle, packed, r, dt, steps, persist=persist, terminal_mask=mask, event_cap=256,
                        actions=np.zeros((steps, 2)))
        assert bits_equal(poses[: o["n_steps"] + 1, r], o["poses"]), r
