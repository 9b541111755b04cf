"""Seeded synthetic scenario batches: the bench / parity workload.

Per scenario: a shared grid of K knot times uniform on [0, L]; every entity drives a constant-speed
arc (start ~ U[-100,100]^2, heading ~ U[-pi,pi], speed ~ U[2,12] m/s, curvature ~ N(0, 0.02) rad/m)
sampled at the grid; bounding box = the reference's car1 catalog entry (2.0 x 4.2 m, centre
(1.37, 0)); `static_frac` of the non-ego entities are static (one knot) and `vanish_frac` only exist
on a strict sub-interval of [0, L].  Scenario r depends only on (seed, r // CHUNK), so any rank can
generate exactly its own shard.
"""
import numpy as np

from . import _lib as L
from .engine import DEFAULT_CTRL, PackedScenarios

SEED = 20240807
CHUNK = 64
CAR1_BBOX = (2.0, 4.2, 1.37, 0.0)  # width, length, center_x, center_y


def _chunk(seed, chunk_id, n, E, K, length, ego_kind, static_frac, vanish_frac, extent):
    rng = np.random.default_rng([seed, chunk_id])
    grid = np.linspace(0.0, length, K)
    x0 = rng.uniform(-extent, extent, (n, E))
    y0 = rng.uniform(-extent, extent, (n, E))
    h0 = rng.uniform(-np.pi, np.pi, (n, E))
    v = rng.uniform(2.0, 12.0, (n, E))
    kappa = rng.normal(0.0, 0.02, (n, E))
    kappa = np.where(np.abs(kappa) < 1e-6, 1e-6, kappa)
    u = rng.random((n, E))
    u[:, 0] = 1.0  # the ego always spans the whole scenario
    static = u < static_frac
    vanish = (~static) & (u < static_frac + vanish_frac)
    # knot index range [a, b] per entity
    a = np.zeros((n, E), np.int64)
    b = np.full((n, E), K - 1, np.int64)
    va = rng.integers(1, K // 2, (n, E))
    vb = rng.integers(K // 2 + 1, K - 1, (n, E))
    a = np.where(vanish, va, a)
    b = np.where(vanish, vb, b)
    ks = rng.integers(0, K, (n, E))
    a = np.where(static, ks, a)
    b = np.where(static, ks, b)
    cnt = (b - a + 1).ravel()
    off = np.concatenate([[0], np.cumsum(cnt)])
    # flat (entity, knot) index lists
    ent = np.repeat(np.arange(n * E), cnt)
    kidx = np.arange(off[-1]) - np.repeat(off[:-1], cnt) + np.repeat(a.ravel(), cnt)
    t = grid[kidx]
    h = h0.ravel()[ent] + kappa.ravel()[ent] * v.ravel()[ent] * t
    x = x0.ravel()[ent] + (np.sin(h) - np.sin(h0.ravel()[ent])) / kappa.ravel()[ent]
    y = y0.ravel()[ent] - (np.cos(h) - np.cos(h0.ravel()[ent])) / kappa.ravel()[ent]
    knots = np.zeros((off[-1], 7))
    knots[:, 0], knots[:, 1], knots[:, 2], knots[:, 4] = t, x, y, h
    kind = np.full((n, E), L.KIND_REPLAY, np.int32)
    kind[:, 0] = ego_kind
    return kind.ravel(), off, knots


def make_batch(n_scenarios, n_entities, n_steps=10000, timestep=1.0 / 30.0, n_knots=128,
               ego_kind=L.KIND_AGENT_REPLAY, static_frac=0.1, vanish_frac=0.1, extent=100.0,
               seed=SEED, first_scenario=0) -> PackedScenarios:
    """Scenarios [first_scenario, first_scenario + n_scenarios) of the seeded synthetic family."""
    R, E = int(n_scenarios), int(n_entities)
    length = n_steps * timestep
    assert first_scenario % CHUNK == 0, "shards start on a chunk boundary"
    kinds, offs, knotss = [], [], []
    rows = 0
    for c0 in range(0, R, CHUNK):
        n = min(CHUNK, R - c0)
        kind, off, knots = _chunk(seed, (first_scenario + c0) // CHUNK, CHUNK, E, n_knots, length,
                                  ego_kind, static_frac, vanish_frac, extent)
        m = n * E
        kinds.append(kind[:m])
        offs.append(off[:m] + rows)
        knotss.append(knots[: off[m]])
        rows += int(off[m])
    knot_off = np.concatenate(offs + [[rows]]).astype(np.int64)
    knots = np.concatenate(knotss, axis=0)
    bbox = np.tile(np.array(CAR1_BBOX), (R * E, 1))
    ego_rows = knot_off[np.arange(R) * E]
    t0 = np.maximum(0.0, knots[ego_rows, 0])  # ScenarioGym.get_start_time
    # Scenario.length = max over entities of the last knot time
    last = knots[knot_off[1:] - 1, 0].reshape(R, E)
    return PackedScenarios(
        R, E, np.concatenate(kinds), np.zeros(R * E, np.int32), bbox, knot_off, knots,
        np.zeros(R, np.int32), t0, last.max(axis=1), np.tile(DEFAULT_CTRL, (R * E, 1)),
    ).validate()


def make_actions(n_steps, n_scenarios, seed=SEED, first_scenario=0):
    """Seeded external (accel, steer) sequences: accel ~ U[-5,5], steer ~ U[-0.7,0.7]; [n, R, 2]."""
    out = np.empty((n_steps, n_scenarios, 2))
    for c0 in range(0, n_scenarios, CHUNK):
        n = min(CHUNK, n_scenarios - c0)
        rng = np.random.default_rng([seed, 7, (first_scenario + c0) // CHUNK])
        a = np.stack([rng.uniform(-5, 5, (n_steps, CHUNK)), rng.uniform(-0.7, 0.7, (n_steps, CHUNK))], -1)
        out[:, c0:c0 + n] = a[:, :n]
    return out


PEDESTRIAN1_BBOX = (0.69, 0.7, 0.0, 0.0)  # width, length, center_x, center_y of the reference's pedestrian1 entry


def make_crowd(n_scenarios, n_entities=256, n_steps=10000, timestep=1.0 / 30.0, side=40.0, radius=3.0,
               seed=SEED, first_scenario=0) -> PackedScenarios:
    """BASELINE config 5: every entity is a PedestrianAgent with the social force model.

    Starts ~ U on a side x side square, two-waypoint routes from the start to a random point on the
    opposite half, desired speed ~ U[0.5, 1.5] * 1.3, SocialForceParameters defaults with the noise off,
    neighbour radius passed explicitly, empty road network, pedestrian1 bounding boxes."""
    R, E = int(n_scenarios), int(n_entities)
    length = n_steps * timestep
    assert first_scenario % CHUNK == 0
    kn, routes, ctrl = [], [], []
    for c0 in range(0, R, CHUNK):
        n = min(CHUNK, R - c0)
        rng = np.random.default_rng([seed, 5, (first_scenario + c0) // CHUNK])
        start = rng.uniform(-side / 2, side / 2, (CHUNK, E, 2))
        goal = -start * rng.uniform(0.3, 1.0, (CHUNK, E, 1)) + rng.normal(0, 2.0, (CHUNK, E, 2))
        h0 = rng.uniform(-np.pi, np.pi, (CHUNK, E))
        vdes = rng.uniform(0.5, 1.5, (CHUNK, E)) * 1.3
        k = np.zeros((CHUNK, E, 2, 7))
        k[:, :, 1, 0] = length
        k[:, :, :, 1:3] = start[:, :, None, :]
        k[:, :, :, 4] = h0[:, :, None]
        kn.append(k[:n].reshape(-1, 7))
        routes.append(np.stack([start, goal], axis=2)[:n].reshape(-1, 2))
        row = np.tile(DEFAULT_CTRL, (n * E, 1))
        row[:, L.C_PED_SPEED_DESIRED] = vdes[:n].ravel()
        row[:, L.C_PED_RADIUS] = radius
        ctrl.append(row)
    return PackedScenarios(
        R, E, np.full(R * E, L.KIND_AGENT_PEDESTRIAN, np.int32), np.ones(R * E, np.int32),
        np.tile(np.array(PEDESTRIAN1_BBOX), (R * E, 1)), np.arange(R * E + 1, dtype=np.int64) * 2,
        np.concatenate(kn), np.zeros(R, np.int32), np.zeros(R), np.full(R, length), np.concatenate(ctrl),
        route_off=np.arange(R * E + 1, dtype=np.int64) * 2, routes=np.concatenate(routes),
    ).validate()


def building_blocks(side=40.0, blocks=2, building=8.0, margin=3.0):
    """A road network in the spirit of examples/crowds.py:149-205 for the crowd workloads: one pavement that covers the whole
    square (walkable) and blocks x blocks square buildings (impenetrable: the nearest-point boundary force of
    social_force.py:190-211) with streets between and around them.  polygon_arrays() layout: ring_off, vert_off, verts, layers."""
    from .road_network import LAYER_IMPENETRABLE, LAYER_PAVEMENT, LAYER_WALKABLE

    h = side / 2 + margin
    rings = [np.array([[-h, -h], [-h, h], [h, h], [h, -h]], np.float64)]
    layers = [LAYER_WALKABLE | LAYER_PAVEMENT]
    pitch = side / blocks
    for i in range(blocks):
        for j in range(blocks):
            cx, cy = -side / 2 + (i + 0.5) * pitch, -side / 2 + (j + 0.5) * pitch
            b = building / 2
            rings.append(np.array([[cx - b, cy - b], [cx - b, cy + b], [cx + b, cy + b], [cx + b, cy - b]], np.float64))
            layers.append(LAYER_IMPENETRABLE | LAYER_WALKABLE)
    return dict(ring_off=np.arange(len(rings) + 1, dtype=np.int64), vert_off=np.arange(len(rings) + 1, dtype=np.int64) * 4,
                verts=np.concatenate(rings), layers=np.array(layers, np.uint32))


def make_crowd_roads(n_scenarios, n_entities=256, n_steps=10000, timestep=1.0 / 30.0, side=40.0, radius=3.0, blocks=2,
                     building=8.0, seed=SEED, first_scenario=0):
    """The config-5 crowd on building_blocks(): every pedestrian starts on a street (never inside a building) and walks a
    route along the streets -- up to four waypoints: along its own street to a crossing, along the crossing street, into the
    goal's street -- as the pavement-graph routes of examples/crowds.py do.  Returns (packed, network, net_of_scenario)."""
    R, E = int(n_scenarios), int(n_entities)
    length = n_steps * timestep
    assert first_scenario % CHUNK == 0
    pitch = side / blocks
    lines = -side / 2 + pitch * np.arange(blocks + 1)      # street centre lines, both directions
    half = (pitch - building) / 2 - 0.6                     # lateral room on a street (a margin to the walls)
    NW = 4
    kn, routes, ctrl = [], [], []
    for c0 in range(0, R, CHUNK):
        n = min(CHUNK, R - c0)
        rng = np.random.default_rng([seed, 7, (first_scenario + c0) // CHUNK])

        def street_points(shape):
            """random points on the streets: (xy, is the street parallel to x, index of its line)"""
            horiz = rng.random(shape) < 0.5
            k = rng.integers(0, blocks + 1, shape)
            along = rng.uniform(-side / 2, side / 2, shape)
            across = lines[k] + rng.uniform(-half, half, shape)
            xy = np.where(horiz[..., None], np.stack([along, across], -1), np.stack([across, along], -1))
            return xy, horiz, k

        start, sh, sk = street_points((CHUNK, E))
        goal, gh, gk = street_points((CHUNK, E))
        via = lines[rng.integers(0, blocks + 1, (CHUNK, E))]  # a crossing street, for start and goal on parallel streets
        wp = np.empty((CHUNK, E, NW, 2))
        wp[:, :, 0] = start
        wp[:, :, 3] = goal
        # perpendicular streets: one corner (the crossing of the two lines), repeated; parallel ones: over the `via` street
        sx, sy, gx, gy = start[..., 0], start[..., 1], goal[..., 0], goal[..., 1]
        perp = sh != gh
        c1 = np.where(sh[..., None], np.stack([np.where(perp, lines[gk], via), sy], -1), np.stack([sx, np.where(perp, lines[gk], via)], -1))
        c2 = np.where(gh[..., None], np.stack([np.where(perp, lines[sk], via), gy], -1), np.stack([gx, np.where(perp, lines[sk], via)], -1))
        wp[:, :, 1] = c1
        wp[:, :, 2] = np.where(perp[..., None], c1, c2)
        wp[:, :, 1:3] += rng.uniform(-0.5, 0.5, (CHUNK, E, 2, 2))  # (nobody aims at the exact same corner point)
        h0 = rng.uniform(-np.pi, np.pi, (CHUNK, E))
        vdes = rng.uniform(0.5, 1.5, (CHUNK, E)) * 1.3
        k = np.zeros((CHUNK, E, 2, 7))
        k[:, :, 1, 0] = length
        k[:, :, :, 1:3] = start[:, :, None, :]
        k[:, :, :, 4] = h0[:, :, None]
        kn.append(k[:n].reshape(-1, 7))
        routes.append(wp[:n].reshape(-1, 2))
        row = np.tile(DEFAULT_CTRL, (n * E, 1))
        row[:, L.C_PED_SPEED_DESIRED] = vdes[:n].ravel()
        row[:, L.C_PED_RADIUS] = radius
        ctrl.append(row)
    packed = PackedScenarios(
        R, E, np.full(R * E, L.KIND_AGENT_PEDESTRIAN, np.int32), np.ones(R * E, np.int32),
        np.tile(np.array(PEDESTRIAN1_BBOX), (R * E, 1)), np.arange(R * E + 1, dtype=np.int64) * 2,
        np.concatenate(kn), np.zeros(R, np.int32), np.zeros(R), np.full(R, length), np.concatenate(ctrl),
        route_off=np.arange(R * E + 1, dtype=np.int64) * NW, routes=np.concatenate(routes),
    ).validate()
    return packed, building_blocks(side, blocks, building), np.zeros(R, np.int32)


def make_crowd_with_car(n_scenarios, n_entities=256, n_steps=10000, timestep=1.0 / 30.0, side=40.0, radius=3.0,
                        seed=SEED, first_scenario=0, car_kind=L.KIND_AGENT_PID) -> PackedScenarios:
    """make_crowd with entity 0 of every scenario replaced by a car (car1, a PIDAgent by default) that crosses the square
    on a straight line while the crowd walks: the all-pedestrian kernel no longer applies, the general pedestrian variant
    runs (examples/crowds.py:149-205 has vehicles among its pedestrians too)."""
    p = make_crowd(n_scenarios, n_entities, n_steps, timestep, side, radius, seed, first_scenario)
    R, E = p.n_scenarios, p.n_entities
    rng = np.random.default_rng([seed, 6, first_scenario // CHUNK])
    length = n_steps * timestep
    cars = np.arange(R) * E
    y0 = rng.uniform(-side / 4, side / 4, R)
    kn = p.knots.reshape(R * E, 2, 7)
    kn[cars, 0, 1], kn[cars, 1, 1] = -side / 2 - 5.0, side / 2 + 5.0
    kn[cars, :, 2] = y0[:, None]
    kn[cars, :, 4] = 0.0
    p.kind[cars] = car_kind
    p.etype[cars] = 0
    p.bbox[cars] = CAR1_BBOX
    p.ctrl[cars] = DEFAULT_CTRL
    # the cars have no route: their two waypoint rows leave the ragged array
    keep = np.ones(R * E, bool)
    keep[cars] = False
    p.routes = p.routes.reshape(R * E, 2, 2)[keep].reshape(-1, 2)
    p.route_off = np.concatenate([[0], np.cumsum(np.where(keep, 2, 0))]).astype(np.int64)
    return p.validate()
