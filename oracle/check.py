"""Final-state check of a device rollout against the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Used by tests/, __graft_entry__.smoke() and bench.py's `verified` leg -- always after the timed region, always as the
checker, never as the thing measured.  Compares, bit for bit, what ScenarioGym exposes after a rollout
(scenario_gym/state/state.py:165-239, scenario_gym.py:308-319): the step count and the clock, every entity's final pose /
velocity / travelled distance / collision row, the ego metric rows and the CollisionMetric event list.
"""
import concurrent.futures as cf

import numpy as np

from . import oracle as O


def _bits(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    na, nb = np.isnan(a), np.isnan(b)
    return a.shape == b.shape and bool(np.array_equal(na, nb) and np.array_equal(a[~na], b[~nb]))


def spread(R, K):
    """K scenario indices spread over [0, R): both ends and evenly between (deterministic)."""
    K = max(1, min(int(K), int(R)))
    return sorted({int(round(i * (R - 1) / max(K - 1, 1))) for i in range(K)})


def oracle_final(packed, r, dt, T, persist=False, terminal_mask=O.TERM_MAX_LENGTH, event_cap=64, sf=None, noise=None,
                 record="last", behaviour="social_force", models=None, model_of=None, road=None):
    from scenario_gym_amd.packing import unpack_scenario

    s = unpack_scenario(packed, r)
    # sg_rollout feeds external-action slots (0, 0) (sgym.h); the oracle wants the rows
    actions = np.zeros((max(int(T), 1), 2)) if (np.asarray(s["kind"]) == O.KIND_AGENT_VEHICLE).any() else None
    return O.rollout(s["knot_off"], s["knots"], s["bbox"], s["etype"], s["kind"], s["ego"], s["t0"], s["length"], dt,
                     persist=persist, terminal_mask=terminal_mask, ctrl=s["ctrl"], actions=actions, max_steps=T, record=record,
                     event_cap=event_cap, route_off=s.get("route_off"), routes=s.get("routes"), sf=sf, noise=noise,
                     behaviour=behaviour, models=models, model_of=model_of, road=road)


def compare_final(st, rows, events, r, o, E, event_cap=64, ped=False, kind=None):
    """Mismatching fields of scenario r (device state / metric rows / events) against the oracle result `o`: [] = equal."""
    bad = []
    E = int(o["poses"].shape[1])  # (a ragged batch: the scenario's own entities, the padding slots are not the oracle's)
    if int(rows["n_steps"][r]) != int(o["n_steps"]):
        bad.append(f"n_steps {rows['n_steps'][r]} != {o['n_steps']}")
    if rows["final_t"][r] != o["final_t"] or st["t"][r] != o["final_t"]:
        bad.append("final_t")
    if bool(rows["done"][r]) != bool(o["is_done"]):
        bad.append("done")
    for k in ("poses", "vels", "dists"):
        if not _bits(st[k][r, :E], o[k][-1]):
            bad.append(k)
    W = o["coll"].shape[-1]
    if not np.array_equal(np.asarray(st["coll"][r]).reshape(len(st["poses"][r]), -1)[:E, :W], o["coll"][-1].reshape(E, W)):
        bad.append("coll")
    # pedestrian lanes: the social force they were moved by; every other lane: the controller state
    is_ped = np.full(E, bool(ped)) if kind is None else (np.asarray(kind)[:E] == O.KIND_AGENT_PEDESTRIAN)
    if is_ped.any() and not _bits(st["force"][r, :E][is_ped], o["extra"][-1][:, 2:][is_ped]):
        bad.append("force")
    if (~is_ped).any() and not _bits(st["ctrl_state"][r, :E][~is_ped], o["extra"][-1][~is_ped]):
        bad.append("ctrl_state")
    for k in ("ego_avg_speed", "ego_max_speed", "ego_distance_travelled"):
        a, b = rows[k][r], o["metric_" + k]
        if not (a == b or (np.isnan(a) and np.isnan(b))):
            bad.append(k)
    if int(rows["n_collisions"][r]) != int(o["n_events"]):
        bad.append(f"n_collisions {rows['n_collisions'][r]} != {o['n_events']}")
    ev = events[events["scenario"] == r]
    m = min(int(o["n_events"]), int(event_cap))
    if len(ev) != m or not (np.array_equal(ev["t"], o["ev_t"][:m]) and np.array_equal(ev["other"], o["ev_other"][:m])
                            and np.array_equal(ev["type"], o["ev_type"][:m])):
        bad.append("events")
    return bad


def verify_engine(eng, packed, dt, T, K=16, event_cap=64, ped=False, rss=False, sf=None, noise_of=None, threads=8,
                  persist=False, terminal_mask=O.TERM_MAX_LENGTH, behaviour="social_force", road_of=None):
    """The engine's CURRENT state (after a rollout of at most T steps from reset) against the oracle on K scenarios spread
    over the batch, full horizon.  noise_of(r) -> the oracle's noise dict of scenario r; road_of(r) -> the polygon arrays of its
    road network (None: none).  rss: also the RSSDistances records
    (codes, safe distances, the two metric flags)."""
    O.build()
    O.lib()
    idx = spread(packed.n_scenarios, K)
    E = packed.n_entities
    st = eng.state()
    rows, events = eng.metrics()
    rs = eng.rss() if rss else None

    def one(r):
        o = oracle_final(packed, r, dt, T, persist=persist, terminal_mask=terminal_mask, event_cap=max(event_cap, 1), sf=sf,
                         noise=None if noise_of is None else noise_of(r), record=True if rss else "last", behaviour=behaviour,
                         road=None if road_of is None else road_of(r))
        bad = compare_final(st, rows, events, r, o, E, event_cap=event_cap, ped=ped, kind=packed.kind[r * E:(r + 1) * E])
        if rss:
            from scenario_gym_amd.packing import unpack_scenario

            s = unpack_scenario(packed, r)
            q = O.rss_rollout(o, s["bbox"], s["ego"])
            codes, safes = q["code"][-1], q["safe"][-1]  # the records of the last executed step (-1 / NaN: no update)
            got_c, got_s = rs[2][r, :E], rs[3][r, :E]
            have = np.ones(E, bool)
            if bool(rs[0][r]) != bool(q["safe_longitudinal"]) or bool(rs[1][r]) != bool(q["safe_lateral"]):
                bad.append("rss_flags")
            if not np.array_equal(got_c[have], codes[have]):
                bad.append("rss_codes")
            if not _bits(got_s[have], safes[have]):
                bad.append("rss_safe_distances")
        return r, bad

    with cf.ThreadPoolExecutor(max(1, min(threads, len(idx)))) as ex:  # ctypes releases the GIL
        res = list(ex.map(one, idx))
    mism = {int(r): bad for r, bad in res if bad}
    return {"scenarios": len(idx), "indices": idx, "steps": int(T), "equal": not mism, "mismatches": mism,
            "fields": ["n_steps", "final_t", "done", "poses", "vels", "dists", "coll", "force" if ped else "ctrl_state",
                       "ego_avg_speed", "ego_max_speed", "ego_distance_travelled", "n_collisions", "events(t, other, type)"]
                      + (["rss flags / codes / safe distances"] if rss else [])}
