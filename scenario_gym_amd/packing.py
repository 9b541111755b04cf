"""Pack scenarios into the sg_scenarios layout (host side, numpy only)."""
from typing import List, Optional, Sequence

import os

import numpy as np

from . import _lib as L
from .engine import DEFAULT_CTRL, PackedScenarios


def default_kinds(n_entities: int, ego: int) -> np.ndarray:
    """The reference's default create_agent (agent.py:151-169): only the entity with ref "ego"
    gets a ReplayTrajectoryAgent, every other entity goes to the BatchReplayEntity."""
    k = np.full(n_entities, L.KIND_REPLAY, np.int32)
    k[ego] = L.KIND_AGENT_REPLAY
    return k


def pack_arrays(scenarios: Sequence[dict], kinds: Optional[Sequence[np.ndarray]] = None,
                ctrls: Optional[Sequence[Optional[np.ndarray]]] = None,
                n_entities: Optional[int] = None) -> PackedScenarios:
    """Batch scenarios given as plain arrays.

    Each scenario dict holds knot_off [E_r+1], knots [rows,7], bbox [E_r,4], etype [E_r], ego,
    t0, length.  Scenarios with fewer entities than the batch width are padded with SG_KIND_NONE.
    """
    R = len(scenarios)
    E = int(n_entities or max(len(s["etype"]) for s in scenarios))
    kind = np.zeros(R * E, np.int32)
    etype = np.full(R * E, 2, np.int32)
    bbox = np.ones((R * E, 4))
    ctrl = np.tile(DEFAULT_CTRL, (R * E, 1))
    knot_off = np.zeros(R * E + 1, np.int64)
    chunks: List[np.ndarray] = []
    rows = 0
    ego, t0, length = np.zeros(R, np.int32), np.zeros(R), np.zeros(R)
    for r, s in enumerate(scenarios):
        Er = len(s["etype"])
        if Er > E:
            raise ValueError(f"scenario {r} has {Er} entities > batch width {E}")
        off = np.asarray(s["knot_off"], np.int64)
        k = default_kinds(Er, int(s["ego"])) if kinds is None else np.asarray(kinds[r], np.int32)
        kind[r * E:r * E + Er] = k
        etype[r * E:r * E + Er] = s["etype"]
        bbox[r * E:r * E + Er] = s["bbox"]
        if ctrls is not None and ctrls[r] is not None:
            ctrl[r * E:r * E + Er] = ctrls[r]
        knot_off[r * E:r * E + Er + 1] = rows + off
        knot_off[r * E + Er + 1:(r + 1) * E + 1] = rows + off[-1]
        if isinstance(s["knots"], list):  # per-entity arrays (pack_scenarios): they join the batch's one concatenation
            chunks.extend(s["knots"])
        else:
            chunks.append(np.asarray(s["knots"], np.float64)[: off[-1]])
        rows += int(off[-1])
        ego[r], t0[r], length[r] = s["ego"], s["t0"], s["length"]
    knots = np.concatenate(chunks, axis=0) if chunks else np.zeros((0, 7))
    route_off = routes = None
    if any("routes" in s and s["routes"] is not None for s in scenarios):
        route_off = np.zeros(R * E + 1, np.int64)
        rchunks, rrows = [], 0
        for r, s in enumerate(scenarios):
            ro = s.get("route_off")
            for e in range(E):
                if ro is not None and e < len(s["etype"]):
                    n = int(ro[e + 1] - ro[e])
                    if n:
                        rchunks.append(np.asarray(s["routes"], np.float64).reshape(-1, 2)[ro[e]:ro[e + 1]])
                        rrows += n
                route_off[r * E + e + 1] = rrows
        routes = np.concatenate(rchunks, axis=0) if rchunks else np.zeros((0, 2))
    return PackedScenarios(R, E, kind, etype, bbox, knot_off, knots, ego, t0, length, ctrl,
                           route_off=route_off, routes=routes).validate()


def unpack_scenario(packed: PackedScenarios, r: int) -> dict:
    """Scenario r of a batch as plain arrays (inverse of pack_arrays, padding removed)."""
    E = packed.n_entities
    kind = packed.kind[r * E:(r + 1) * E]
    n = int((kind != L.KIND_NONE).sum()) if (kind != L.KIND_NONE).any() else 0
    n = max(n, int(np.max(np.nonzero(kind != L.KIND_NONE)[0])) + 1 if n else 0)
    off = packed.knot_off[r * E:r * E + n + 1]
    return dict(
        knot_off=off - off[0], knots=packed.knots[off[0]:off[-1]], bbox=packed.bbox[r * E:r * E + n],
        etype=packed.etype[r * E:r * E + n], kind=kind[:n], ego=int(packed.ego[r]), t0=float(packed.t0[r]),
        length=float(packed.length[r]),
        ctrl=None if packed.ctrl is None else packed.ctrl[r * E:r * E + n],
        route_off=None if packed.route_off is None else packed.route_off[r * E:r * E + n + 1] - packed.route_off[r * E],
        routes=None if packed.route_off is None else packed.routes[packed.route_off[r * E]:packed.route_off[r * E + n]],
    )


def pack_scenarios(scenarios, create_agent=None):
    """Scenario objects -> PackedScenarios, mirroring ScenarioGym.set_scenario + create_agents
    (reference scenario_gym.py:157-215): entities whose create_agent() returns None replay through the
    batch path, the others become agent lanes of their device kind."""
    from .agent import _create_agent
    from .entity import catalog_type_code

    create_agent = create_agent or _create_agent
    arrays, kinds, ctrls, agents = [], [], [], []
    for sc in scenarios:
        ents = sc.entities
        off = np.concatenate([[0], np.cumsum([len(e.trajectory) for e in ents])]).astype(np.int64)
        ego = ents.index(sc.ego)
        kind = np.full(len(ents), L.KIND_REPLAY, np.int32)
        ctrl = np.tile(DEFAULT_CTRL, (len(ents), 1))
        sc_agents = {}
        for i, e in enumerate(ents):
            agent = create_agent(sc, e)
            if agent is not None:
                kind[i] = agent.device_kind()
                if hasattr(agent, "ctrl_row"):
                    ctrl[i] = agent.ctrl_row()
                elif hasattr(agent.controller, "ctrl_row"):  # caller-run agents may bring any controller object
                    ctrl[i] = agent.controller.ctrl_row()
                sc_agents[e] = agent
        routes = [np.asarray(sc_agents[e].route, np.float64).reshape(-1, 2) if e in sc_agents and hasattr(sc_agents[e], "route")
                  else np.zeros((0, 2)) for e in ents]
        arrays.append(dict(
            route_off=np.concatenate([[0], np.cumsum([len(x) for x in routes])]).astype(np.int64),
            routes=np.concatenate(routes, axis=0) if any(len(x) for x in routes) else None,
            knot_off=off, knots=[e.trajectory.data for e in ents],
            bbox=np.array([[e.bounding_box.width, e.bounding_box.length, e.bounding_box.center_x,
                            e.bounding_box.center_y] for e in ents], np.float64),
            etype=np.array([catalog_type_code(e) for e in ents], np.int32), ego=ego,
            t0=max(0.0, float(sc.ego.trajectory.min_t)),  # ScenarioGym.get_start_time
            length=float(sc.length),
        ))
        kinds.append(kind)
        ctrls.append(ctrl)
        agents.append(sc_agents)
    packed = pack_arrays(arrays, kinds=kinds, ctrls=ctrls)
    packed.refs = [[e.ref for e in sc.entities] for sc in scenarios]
    return packed, agents


def _file_arrays(path: str, relabel: bool):
    """One OpenSCENARIO file as the arrays pack_arrays takes -- what import_scenario + pack_scenarios produce for it with the
    reference's default agents (the ego replays its trajectory as an agent, everybody else through the batch path), without a
    Scenario / Entity / Trajectory / Agent object in between: the native scan, the catalog boxes, the last trajectory
    assignment per entity (read.py:133-217), Trajectory.__init__'s normalisation (trajectory.py:34-96), relabel
    (read.py:244-273).  Returns None for anything off that path (a road network whose elevation would be needed, inline
    entity definitions, files the strict scan refuses, an entity without a trajectory): the caller takes the object route."""
    from . import xosc as X
    from .entity import catalog_type_code
    from .trajectory import Trajectory

    with open(path, "rb") as f:
        text = f.read()
    try:
        scan = X.scan_xosc(text)
    except (ValueError, UnicodeDecodeError):
        return None
    if any(o["catalog"] is None for o in scan["objects"]):
        return None
    cwd = os.path.dirname(path)
    catalogs = {}
    for d in scan["dirs"]:
        d = d if os.path.isabs(d) else os.path.join(cwd, d)
        for fn in X.catalog_files(d):
            name, entries = X.read_catalog(fn)
            catalogs[name] = entries
    protos = {}
    for o in scan["objects"]:
        try:
            protos[o["name"]] = catalogs[o["catalog"]][o["entry"]]
        except KeyError:
            return None  # (import_scenario warns and drops the entity)
    last = {}
    for ref, knot in scan["teleports"]:
        if ref in protos:
            last[ref] = (knot[None, :], False)
    for ref, verts in scan["trajectories"]:
        if ref in protos and len(verts):
            last[ref] = (verts, True)
    if len(last) != len(protos):
        return None
    if scan["road_file"] and any(is_traj and np.isnan(v[:, 3]).any() for v, is_traj in last.values()):
        return None  # read.py:212-215 would fill z from the road network's elevation
    names = list(protos)
    knots = Trajectory.many_arrays([last[n][0] for n in names])
    ents = [protos[n] for n in names]
    if relabel:
        counts = {"vehicle": 0, "pedestrian": 0, "other": 0}
        refs = ["ego"]
        for e in ents[1:]:
            key = "vehicle" if isinstance(e, X.Vehicle) else "pedestrian" if isinstance(e, X.Pedestrian) else "other"
            refs.append(f"{key}_{counts[key]}")
            counts[key] += 1
    else:
        refs = names
    ego = refs.index("ego") if "ego" in refs else 0  # Scenario.ego: the entity called "ego", else the first one
    off = np.zeros(len(names) + 1, np.int64)
    np.cumsum([len(k) for k in knots], out=off[1:])
    t_first, t_last = float(knots[ego][0, 0]), float(max(k[-1, 0] for k in knots))
    if isinstance(knots, np.ndarray):  # one block for the file: a single chunk for the batch's concatenation
        knots = knots.reshape(-1, 7)
    kind = np.full(len(names), L.KIND_REPLAY, np.int32)
    if refs[ego] == "ego":
        kind[ego] = L.KIND_AGENT_REPLAY  # _create_agent: only the entity with ref "ego" gets an agent
    return dict(knot_off=off, knots=knots, ego=ego, t0=max(0.0, t_first), length=t_last,
                bbox=np.array([[e.bounding_box.width, e.bounding_box.length, e.bounding_box.center_x, e.bounding_box.center_y]
                               for e in ents], np.float64),
                etype=np.array([catalog_type_code(e) for e in ents], np.int32)), kind, refs


def load_and_pack(paths: Sequence[str], n_entities: Optional[int] = None, relabel: bool = True) -> PackedScenarios:
    """Import the files and pack them with the reference's default agents: what a worker process of a many-file sweep
    returns -- a few large arrays instead of thousands of Python objects.  OpenSCENARIO files on the usual path never become
    objects at all (_file_arrays); the others go through xosc.load_scenario_file + pack_scenarios.  Same arrays either way
    (tests/test_ingest_json.py::test_bulk_ingest_equals_object_path)."""
    from .xosc import load_scenario_file

    arrays, kinds, refs = [], [], []
    for p in paths:
        got = _file_arrays(p, relabel) if (os.path.splitext(p)[1].lower() != ".json" and os.environ.get("SG_INGEST_OBJECTS") != "1") else None
        if got is None:
            one, _ = pack_scenarios([load_scenario_file(p, relabel=relabel)])
            E1 = one.n_entities
            n = int((one.kind != L.KIND_NONE).sum())
            got = (dict(knot_off=one.knot_off[: n + 1], knots=one.knots, ego=int(one.ego[0]), t0=float(one.t0[0]),
                        length=float(one.length[0]), bbox=one.bbox[:n], etype=one.etype[:n]), one.kind[:n], one.refs[0])
            assert n == E1
        arrays.append(got[0])
        kinds.append(got[1])
        refs.append(got[2])
    packed = pack_arrays(arrays, kinds=kinds)
    packed.refs = refs
    if n_entities is not None and n_entities != packed.n_entities:
        packed = widen_packed(packed, n_entities)
    return packed


def widen_packed(p: PackedScenarios, E: int) -> PackedScenarios:
    """The same batch with E >= p.n_entities entity slots per scenario (padding slots: kind NONE)."""
    R, E0 = p.n_scenarios, p.n_entities
    if E < E0:
        raise ValueError(f"batch has {E0} entities per scenario > {E}")

    def pad(a, fill):
        out = np.full((R, E) + a.shape[1:], fill, a.dtype)
        out[:, :E0] = a.reshape((R, E0) + a.shape[1:])
        return out.reshape((R * E,) + a.shape[1:])

    off = p.knot_off[:-1].reshape(R, E0)
    end = p.knot_off[1:].reshape(R, E0)[:, -1]                 # rows of a scenario's padding slots: empty, at its end
    ko = np.concatenate([np.concatenate([off, np.repeat(end[:, None], E - E0, axis=1)], axis=1).ravel(), p.knot_off[-1:]])
    ctrl = None
    if p.ctrl is not None:
        ctrl = np.tile(DEFAULT_CTRL, (R * E, 1)).reshape(R, E, -1)
        ctrl[:, :E0] = p.ctrl.reshape(R, E0, -1)
        ctrl = ctrl.reshape(R * E, -1)
    if p.route_off is not None:
        raise NotImplementedError("widen_packed: batches with pedestrian routes")
    out = PackedScenarios(R, E, pad(p.kind, 0), pad(p.etype, 2), pad(p.bbox, 1.0), ko, p.knots, p.ego, p.t0, p.length, ctrl)
    out.refs = p.refs
    return out.validate()


def merge_packed(parts: Sequence[PackedScenarios]) -> PackedScenarios:
    """Batches of the same entity width, one after the other, as one batch."""
    E = parts[0].n_entities
    if any(q.n_entities != E for q in parts) or any(q.route_off is not None for q in parts):
        raise ValueError("merge_packed: same n_entities, no pedestrian routes")
    rows = np.cumsum([0] + [len(q.knots) for q in parts])
    ko = np.concatenate([q.knot_off[:-1] + r for q, r in zip(parts, rows)] + [rows[-1:]])
    cat = lambda f: np.concatenate([getattr(q, f) for q in parts])  # noqa: E731
    out = PackedScenarios(sum(q.n_scenarios for q in parts), E, cat("kind"), cat("etype"), cat("bbox"), ko.astype(np.int64),
                          cat("knots"), cat("ego"), cat("t0"), cat("length"),
                          None if parts[0].ctrl is None else cat("ctrl"))
    out.refs = [r for q in parts for r in q.refs]
    return out.validate()


def packed_to_shm(p: PackedScenarios) -> dict:
    """A packed batch handed from a worker process to its parent WITHOUT pickling its knots (the bulk of it: tens of MB per
    task of a many-file sweep, a gigabyte per few thousand files -- the parent's single result thread unpickles them one
    after the other and becomes the bottleneck of the whole ingest): the knot rows go into a shared-memory segment, the
    small arrays travel as usual.  The parent calls packed_from_shm, which copies them out and removes the segment."""
    from multiprocessing import shared_memory

    kn = np.ascontiguousarray(p.knots, np.float64)
    small = dict(shape=kn.shape, R=p.n_scenarios, E=p.n_entities, kind=p.kind, etype=p.etype, bbox=p.bbox,
                 knot_off=p.knot_off, ego=p.ego, t0=p.t0, length=p.length, ctrl=p.ctrl, refs=getattr(p, "refs", None))
    # A /dev/shm that cannot hold the rows (containers default to 64 MB) would kill this process with SIGBUS at the first
    # page it cannot back, not raise: ask first, and send the rows through the ordinary pickle of the result otherwise.
    try:
        st = os.statvfs("/dev/shm")
        room = st.f_bavail * st.f_frsize >= 2 * kn.nbytes + (16 << 20)
    except OSError:
        room = False
    if not room:
        return dict(small, shm=None, knots=kn)
    try:
        shm = shared_memory.SharedMemory(create=True, size=max(kn.nbytes, 8))
    except OSError:
        return dict(small, shm=None, knots=kn)
    np.ndarray(kn.shape, np.float64, buffer=shm.buf)[...] = kn
    meta = dict(shm=shm.name, shape=kn.shape, R=p.n_scenarios, E=p.n_entities, kind=p.kind, etype=p.etype, bbox=p.bbox,
                knot_off=p.knot_off, ego=p.ego, t0=p.t0, length=p.length, ctrl=p.ctrl, refs=getattr(p, "refs", None))
    shm.close()
    try:  # the parent unlinks it: keep this process's resource tracker from doing so at exit
        from multiprocessing import resource_tracker

        resource_tracker.unregister(shm._name, "shared_memory")
    except Exception:
        pass
    return meta


def packed_from_shm(meta: dict) -> PackedScenarios:
    from multiprocessing import shared_memory

    if meta["shm"] is None:
        knots = meta["knots"]
    else:
        shm = shared_memory.SharedMemory(name=meta["shm"])
        try:
            knots = np.ndarray(meta["shape"], np.float64, buffer=shm.buf).copy()
        finally:
            shm.close()
            shm.unlink()
    out = PackedScenarios(meta["R"], meta["E"], meta["kind"], meta["etype"], meta["bbox"], meta["knot_off"], knots, meta["ego"],
                          meta["t0"], meta["length"], meta["ctrl"])
    out.refs = meta["refs"]
    return out


def release_shm(metas) -> int:
    """Remove the segments of results that were received and will not be merged (the parent failed in between: a merge
    error, KeyboardInterrupt, a broken pool).  Segments that are gone already are skipped; returns how many it removed."""
    from multiprocessing import shared_memory

    n = 0
    for m in metas:
        name = m.get("shm") if isinstance(m, dict) else None
        if not name:
            continue
        try:
            shm = shared_memory.SharedMemory(name=name)
        except (FileNotFoundError, OSError):
            continue
        shm.close()
        try:
            shm.unlink()
            n += 1
        except FileNotFoundError:
            pass
    return n


def merge_packed_shm(metas: Sequence[dict], threads: int = 4) -> PackedScenarios:
    """merge_packed([packed_from_shm(m) for m in metas]) with ONE copy of the knots: every worker's rows go from its
    shared-memory segment straight into their slice of the merged array."""
    from multiprocessing import shared_memory

    E = metas[0]["E"]
    if any(m["E"] != E for m in metas):
        raise ValueError("merge_packed_shm: same n_entities")
    rows = np.cumsum([0] + [m["shape"][0] for m in metas])
    knots = np.empty((int(rows[-1]), 7))

    def one(args):  # (numpy's copy releases the GIL: the segments are copied -- and their pages faulted in -- side by side)
        m, a, b = args
        if m["shm"] is None:
            knots[a:b] = m["knots"]
            return
        shm = shared_memory.SharedMemory(name=m["shm"])
        try:
            knots[a:b] = np.ndarray(m["shape"], np.float64, buffer=shm.buf)
        finally:
            shm.close()
            shm.unlink()

    jobs = list(zip(metas, rows[:-1], rows[1:]))
    try:
        if threads > 1 and len(jobs) > 1:
            from concurrent.futures import ThreadPoolExecutor

            with ThreadPoolExecutor(threads) as ex:
                list(ex.map(one, jobs))
        else:
            for j in jobs:
                one(j)
    except BaseException:
        release_shm(metas)  # whatever was not copied yet
        raise
    ko = np.concatenate([m["knot_off"][:-1] + r for m, r in zip(metas, rows)] + [rows[-1:]])
    cat = lambda f: np.concatenate([m[f] for m in metas])  # noqa: E731
    out = PackedScenarios(sum(m["R"] for m in metas), E, cat("kind"), cat("etype"), cat("bbox"), ko.astype(np.int64), knots,
                          cat("ego"), cat("t0"), cat("length"), None if metas[0]["ctrl"] is None else cat("ctrl"))
    out.refs = [r for m in metas for r in (m["refs"] or [])]
    return out.validate()


def effective_cpus() -> int:
    """Host CPUs this process may actually use: os.cpu_count() capped by the cgroup CPU quota (a box can show 256 logical
    CPUs under a quota of 16: more busy threads or processes than that are throttled, not run)."""
    import os

    n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:  # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return n
