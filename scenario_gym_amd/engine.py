"""RolloutEngine: thin object wrapper over one sg_handle (one GPU, R scenarios x E entity slots)."""
import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import _lib as L

TERMINAL_BITS = {"max_length": L.TERM_MAX_LENGTH, "collision": L.TERM_COLLISION,
                 "ego_collision": L.TERM_EGO_COLLISION, "ego_off_road": L.TERM_EGO_OFF_ROAD}

# VehicleController / PIDController constructor defaults (reference controller.py:64-70, 157-161)
SCEN_DTYPE = np.dtype([  # sg_scenario_state
    ("t", "f8"), ("prev_t", "f8"), ("ego_avg_speed", "f8"), ("ego_max_speed", "f8"), ("avg_t", "f8"),
    ("ego_distance_travelled", "f8"), ("last_row", "u8", (4,)), ("done", "i4"), ("n_steps", "i4"),
    ("n_events", "i4"), ("rec_rows", "i4"), ("noise_pos", "i8"), ("last_row_hi", "u8", (4,))])

# + PedestrianAgent / PedestrianController defaults (pedestrian/agent.py:18-27): speed_desired (set per agent),
# max_speed 5.0, head_rot_angle 0.0, distance_threshold 1.0
DEFAULT_CTRL = np.array([0.7, 5.0, np.nan, 0.0, 0.03054, 1.5709, 0.3753, 1.8970, 0.0204, 0.0, 5.0, 0.0, 1.0, 0, 0, 0])


@dataclass
class PackedScenarios:
    """Numeric content of R scenarios x E entity slots in the sg_scenarios layout."""

    n_scenarios: int
    n_entities: int
    kind: np.ndarray      # [R*E] int32
    etype: np.ndarray     # [R*E] int32
    bbox: np.ndarray      # [R*E, 4]
    knot_off: np.ndarray  # [R*E+1] int64
    knots: np.ndarray     # [rows, 7]
    ego: np.ndarray       # [R] int32
    t0: np.ndarray        # [R]
    length: np.ndarray    # [R]
    ctrl: Optional[np.ndarray] = None  # [R*E, 16]
    refs: list = field(default_factory=list)
    route_off: Optional[np.ndarray] = None  # [R*E+1] int64, pedestrian agents' waypoint rows
    routes: Optional[np.ndarray] = None     # [rows, 2]

    def validate(self):
        R, E = self.n_scenarios, self.n_entities
        assert self.kind.shape == (R * E,) and self.etype.shape == (R * E,)
        assert self.bbox.shape == (R * E, 4) and self.knot_off.shape == (R * E + 1,)
        assert self.knots.ndim == 2 and self.knots.shape[1] == 7
        assert self.ego.shape == (R,) and self.t0.shape == (R,) and self.length.shape == (R,)
        assert self.ctrl is None or self.ctrl.shape == (R * E, L.NCTRL)
        assert self.route_off is None or (self.route_off.shape == (R * E + 1,) and self.routes.shape[1] == 2)
        return self

    def pin(self, device=0):
        """Move the knots -- nearly all of the bytes of a batch -- into page-locked host memory (sg_host_alloc), in place:
        sg_upload then copies them at the PCIe rate without a staging copy on the host."""
        pinned = L.pinned_empty(self.knots.shape, np.float64, device)
        pinned[...] = self.knots
        self.knots = pinned
        return self

    def shard(self, lo, hi):
        """Scenarios [lo, hi) as an independent batch (replica sharding across GPUs)."""
        E = self.n_entities
        a, b = int(self.knot_off[lo * E]), int(self.knot_off[hi * E])
        ro = rt = None
        if self.route_off is not None:
            ra, rb = int(self.route_off[lo * E]), int(self.route_off[hi * E])
            ro, rt = self.route_off[lo * E:hi * E + 1] - ra, self.routes[ra:rb]
        return PackedScenarios(
            hi - lo, E, self.kind[lo * E:hi * E], self.etype[lo * E:hi * E], self.bbox[lo * E:hi * E],
            self.knot_off[lo * E:hi * E + 1] - a, self.knots[a:b], self.ego[lo:hi], self.t0[lo:hi],
            self.length[lo:hi], None if self.ctrl is None else self.ctrl[lo * E:hi * E],
            self.refs[lo:hi] if self.refs else [], ro, rt,
        )


def terminal_mask(conditions):
    if conditions is None:
        return L.TERM_MAX_LENGTH
    mask = 0
    for c in conditions:
        if c not in TERMINAL_BITS:
            raise ValueError(f"terminal condition {c!r} is not available on the device path "
                             f"(supported: {sorted(TERMINAL_BITS)})")
        mask |= TERMINAL_BITS[c]
    return mask


class RolloutEngine:
    """ScenarioGym's step loop for a whole batch, resident on one MI355X."""

    def __init__(self, n_scenarios, n_entities, timestep=1.0 / 30.0, persist=False,
                 terminal_conditions=None, record_capacity=0, event_capacity=16, device=0, social_force=None):
        self.lib = L.load()
        self.R, self.E = int(n_scenarios), int(n_entities)
        self.cfg = L.SgConfig(int(device), self.R, self.E, int(bool(persist)),
                              terminal_mask(terminal_conditions), int(record_capacity),
                              int(event_capacity), 0, float(timestep))
        self.h = C.c_void_p()
        rc = self.lib.sg_create(C.byref(self.cfg), C.byref(self.h))
        if rc != L.SG_OK:
            msg = self.lib.sg_last_error(None).decode()
            self.h = None
            raise RuntimeError(f"sg_create failed ({rc}): {msg}")
        self._view = None
        self._keep = None
        if social_force is not None and "models" in social_force:  # per-agent models (BatchedScenarioGym)
            self.set_ped_models(social_force["models"], social_force.get("model_of"), noise=social_force.get("noise"),
                                noise_seed=social_force.get("noise_seed", 0), normals=social_force.get("normals"))
        elif social_force is not None:
            self.set_social_force(**social_force)

    def set_social_force(self, relaxation_time=1.5, ped_repulse_V=1.0, ped_repulse_sigma=1.0, ped_attract_C=0.0,
                         sight_weight=0.5, sight_weight_use=True, sight_angle=200, max_speed_factor=1.3,
                         bias_lon=0.0, bias_lat=0.0, imp_boundary_repulse_U=2.0, imp_boundary_repulse_R=0.1,
                         std_lon=0.0, std_lat=0.0, noise=None, noise_seed=0, normals=None, behaviour="social_force"):
        """SocialForceParameters of every pedestrian agent on this handle (pedestrian/social_force.py:16-30);
        call before upload().  behaviour="random_walk": the pedestrians follow RandomWalk (pedestrian/random_walk.py:22-44)
        instead, which reads bias_lon / bias_lat / std_lon / std_lat only.  std_lon / std_lat with noise="device" (counter-based generator on the GPU, the default when
        a std is non-zero) or noise="stream" + normals[R, n] (the variates numpy's legacy generator would hand out:
        np.random.RandomState(k).standard_normal(n) per scenario) are the random fluctuations of :106-108."""
        sf = L.SgSocialForce(relaxation_time, ped_repulse_V, ped_repulse_sigma, ped_attract_C, sight_weight,
                             float(bool(sight_weight_use)), float(np.cos(sight_angle / 2 * np.pi / 180)),
                             max_speed_factor, bias_lon, bias_lat, imp_boundary_repulse_U, imp_boundary_repulse_R)
        self._check(self.lib.sg_set_social_force(self.h, C.byref(sf)), "sg_set_social_force")
        self.set_ped_behaviour(behaviour)
        if noise is None:
            noise = "device" if (std_lon != 0 or std_lat != 0) else "off"
        self.set_ped_noise(noise, std_lon, std_lat, normals=normals, seed=noise_seed)

    def set_ped_models(self, models, model_of=None, noise=None, noise_seed=0, normals=None):
        """sg_set_ped_models: per-agent behaviour models (pedestrian/agent.py:18-41: every PedestrianAgent holds its own
        behaviour object).  models: dicts of set_social_force()'s parameter names (+ behaviour, std_lon, std_lat);
        model_of[R * E]: the model of every entity slot (ignored where there is no pedestrian agent).  The noise MODE is one
        per handle (noise = "off" / "device" / "stream" + normals; default: "device" when any std is non-zero); before upload()."""
        rows = (L.SgPedModel * len(models))()
        any_std = False
        for i, m in enumerate(models):
            m = dict(m)
            beh = m.pop("behaviour", "social_force")
            std_lon, std_lat = float(m.pop("std_lon", 0.0)), float(m.pop("std_lat", 0.0))
            for k in ("noise", "noise_seed", "normals"):
                m.pop(k, None)
            d = dict(relaxation_time=1.5, ped_repulse_V=1.0, ped_repulse_sigma=1.0, ped_attract_C=0.0, sight_weight=0.5,
                     sight_weight_use=True, sight_angle=200, max_speed_factor=1.3, bias_lon=0.0, bias_lat=0.0,
                     imp_boundary_repulse_U=2.0, imp_boundary_repulse_R=0.1)
            unknown = set(m) - set(d)
            if unknown:
                raise TypeError(f"set_ped_models: unknown parameters {sorted(unknown)}")
            d.update(m)
            rows[i].behaviour = {"social_force": L.PED_SOCIAL_FORCE, "random_walk": L.PED_RANDOM_WALK}[beh]
            rows[i].params = L.SgSocialForce(d["relaxation_time"], d["ped_repulse_V"], d["ped_repulse_sigma"], d["ped_attract_C"],
                                             d["sight_weight"], float(bool(d["sight_weight_use"])),
                                             float(np.cos(d["sight_angle"] / 2 * np.pi / 180)), d["max_speed_factor"], d["bias_lon"],
                                             d["bias_lat"], d["imp_boundary_repulse_U"], d["imp_boundary_repulse_R"])
            rows[i].std_lon, rows[i].std_lat = std_lon, std_lat
            any_std = any_std or std_lon != 0 or std_lat != 0
        if noise is None:
            noise = "device" if any_std else "off"
        # the mode first (its std arguments are those of a one-model handle; the models carry their own)
        self.set_ped_noise(noise, rows[0].std_lon, rows[0].std_lat, normals=normals, seed=noise_seed)
        mo = None
        if model_of is not None:
            mo = np.ascontiguousarray(model_of, np.int32).ravel()
            if mo.size != self.R * self.E:
                raise ValueError(f"model_of must have n_scenarios * n_entities = {self.R * self.E} entries")
        self._check(self.lib.sg_set_ped_models(self.h, len(models), rows, None if mo is None else mo.ctypes.data), "sg_set_ped_models")

    def set_ped_behaviour(self, behaviour="social_force"):
        """sg_set_ped_behaviour: "social_force" or "random_walk" for every pedestrian agent of the handle; before upload()."""
        code = {"social_force": L.PED_SOCIAL_FORCE, "random_walk": L.PED_RANDOM_WALK}[behaviour]
        self._check(self.lib.sg_set_ped_behaviour(self.h, code), "sg_set_ped_behaviour")

    def set_ped_noise(self, mode="off", std_lon=0.0, std_lat=0.0, normals=None, seed=0):
        """sg_set_ped_noise: "off", "stream" (normals[R, n] standard normal variates per scenario) or "device"."""
        code = {"off": L.NOISE_OFF, "stream": L.NOISE_STREAM, "device": L.NOISE_DEVICE}[mode]
        ptr, n = None, 0
        if code == L.NOISE_STREAM:
            normals = np.ascontiguousarray(normals, np.float64)
            if normals.ndim != 2 or normals.shape[0] != self.R:
                raise ValueError(f"normals must be [n_scenarios = {self.R}, n]")
            ptr, n = normals.ctypes.data_as(C.c_void_p), normals.shape[1]
        self._check(self.lib.sg_set_ped_noise(self.h, code, float(std_lon), float(std_lat), ptr, int(n), int(seed) & (2 ** 64 - 1)),
                    "sg_set_ped_noise")

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc, what):
        if rc != L.SG_OK:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.sg_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.sg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ ABI calls
    def upload(self, packed: PackedScenarios):
        packed.validate()
        if (packed.n_scenarios, packed.n_entities) != (self.R, self.E):
            raise ValueError("packed batch does not match the engine's (n_scenarios, n_entities)")
        arrs = dict(
            kind=np.ascontiguousarray(packed.kind, np.int32), etype=np.ascontiguousarray(packed.etype, np.int32),
            bbox=np.ascontiguousarray(packed.bbox, np.float64), knot_off=np.ascontiguousarray(packed.knot_off, np.int64),
            knots=np.ascontiguousarray(packed.knots, np.float64),
            ctrl=None if packed.ctrl is None else np.ascontiguousarray(packed.ctrl, np.float64),
            ego=np.ascontiguousarray(packed.ego, np.int32), t0=np.ascontiguousarray(packed.t0, np.float64),
            length=np.ascontiguousarray(packed.length, np.float64),
            route_off=None if packed.route_off is None else np.ascontiguousarray(packed.route_off, np.int64),
            routes=None if packed.routes is None else np.ascontiguousarray(packed.routes, np.float64),
        )
        sc = L.SgScenarios(*[None if arrs[n] is None else arrs[n].ctypes.data for n, _ in L.SgScenarios._fields_])
        self._check(self.lib.sg_upload(self.h, C.byref(sc)), "sg_upload")
        self._view = L.SgStateView()
        self._check(self.lib.sg_state_view_get(self.h, C.byref(self._view)), "sg_state_view_get")
        return self

    def reset(self):
        self._check(self.lib.sg_reset(self.h), "sg_reset")

    def set_timestep(self, dt):
        self._check(self.lib.sg_set_timestep(self.h, float(dt)), "sg_set_timestep")

    def step(self, n_steps=1, actions=None):
        """n x ScenarioGym.step(); actions [n, R, 2] (accel, steer) for external-action slots."""
        ptr, on_device = None, 0
        if actions is not None and hasattr(actions, "data_ptr") and getattr(actions, "is_cuda", False):
            # a torch tensor on this GPU (fp64, contiguous): its device pointer goes down as it is
            import torch

            assert actions.dtype == torch.float64 and actions.is_contiguous() and actions.numel() == n_steps * self.R * 2
            torch.cuda.current_stream(actions.device).synchronize()  # the policy's writes before the handle's stream reads
            ptr, on_device = actions.data_ptr(), 1
        elif actions is not None:
            actions = np.ascontiguousarray(actions, np.float64).reshape(n_steps, self.R, 2)
            ptr = actions.ctypes.data
        self._check(self.lib.sg_step(self.h, int(n_steps), ptr, on_device), "sg_step")

    def set_external_poses(self, poses):
        """Poses [R, E, 6] of the caller-run agents (KIND_AGENT_EXTERNAL slots; NaN x = the agent returned None)."""
        poses = np.ascontiguousarray(poses, np.float64).reshape(self.R, self.E, 6)
        self._check(self.lib.sg_set_external_poses(self.h, poses.ctypes.data), "sg_set_external_poses")

    def future_collision(self, horizon=5.0, n_samples=10):
        """FutureCollisionDetector (sensor/common.py:59-106) for the ego of every scenario at the current time: bool [R]."""
        out = np.zeros(self.R, np.uint8)
        self._check(self.lib.sg_future_collision(self.h, float(horizon), int(n_samples), out.ctypes.data), "sg_future_collision")
        return out.astype(bool)

    def raster_entities(self, width=20.0, height=20.0, nw=20, nh=20):
        """RasterizedMapSensor "entity" layer (sensor/map.py:120-192) around the ego of every scenario: bool [R, nh, nw]."""
        out = np.zeros((self.R, int(nh), int(nw)), np.uint8)
        self._check(self.lib.sg_raster_entities(self.h, float(width), float(height), int(nw), int(nh), out.ctypes.data),
                    "sg_raster_entities")
        return out.astype(bool)

    def reset_scenarios(self, mask):
        """State.reset for the scenarios with mask[r] != 0; the others keep their state."""
        m = np.ascontiguousarray(mask, np.uint8)
        assert m.shape == (self.R,)
        self._check(self.lib.sg_reset_scenarios(self.h, m.ctypes.data), "sg_reset_scenarios")

    def terminal_flags(self):
        """TERM_* bits of every scenario's current state, all four conditions evaluated: uint32 [R]."""
        out = np.zeros(self.R, np.uint32)
        self._check(self.lib.sg_terminal_flags(self.h, out.ctypes.data, None), "sg_terminal_flags")
        return out

    def set_road_networks(self, networks, net_of_scenario):
        """Scenario.road_network of the uploaded batch.  networks: list of polygon_arrays() dicts (ring_off, vert_off, verts,
        layers; scenario_gym_amd.road_network.RoadNetwork), net_of_scenario: [R] index into it, -1 = no road network.
        Needed by the ego_off_road terminal condition and the surface layers of raster_map; call after upload()."""
        nos = np.ascontiguousarray(net_of_scenario, np.int32)
        assert nos.shape == (self.R,)
        poly_off, ring_off, vert_off, verts, layers = [0], [np.zeros(1, np.int64)], [np.zeros(1, np.int64)], [], []
        for a in networks:
            ro, vo = np.asarray(a["ring_off"], np.int64), np.asarray(a["vert_off"], np.int64)
            ring_off.append(ro[1:] + ring_off[-1][-1])
            vert_off.append(vo[1:] + vert_off[-1][-1])
            verts.append(np.asarray(a["verts"], np.float64).reshape(-1, 2))
            layers.append(np.asarray(a["layers"], np.uint32))
            poly_off.append(poly_off[-1] + len(a["layers"]))
        poly_off = np.array(poly_off, np.int64)
        ring_off, vert_off = np.concatenate(ring_off), np.concatenate(vert_off)
        verts = np.ascontiguousarray(np.concatenate(verts) if verts else np.zeros((0, 2)))
        layers = np.ascontiguousarray(np.concatenate(layers) if layers else np.zeros(0, np.uint32))
        p = lambda a: a.ctypes.data  # noqa: E731
        st = L.SgRoadNetworks(len(networks), p(nos), p(poly_off), p(ring_off), p(vert_off), p(verts), p(layers))
        self._check(self.lib.sg_set_road_networks(self.h, C.byref(st)), "sg_set_road_networks")

    def raster_map(self, layers, width=20.0, height=20.0, nw=20, nh=20):
        """RasterizedMapSensor._step (sensor/map.py:136-149) around the ego of every scenario: bool [R, n_layers, nh, nw];
        layers: 0 = entity, or one LAYER_* bit of scenario_gym_amd.road_network."""
        lay = np.ascontiguousarray(layers, np.int32)
        out = np.zeros((self.R, len(lay), int(nh), int(nw)), np.uint8)
        self._check(self.lib.sg_raster_map(self.h, float(width), float(height), int(nw), int(nh), len(lay), lay.ctypes.data,
                                           out.ctypes.data), "sg_raster_map")
        return out.astype(bool)

    def raster_map_torch(self, layers, width=20.0, height=20.0, nw=20, nh=20):
        """raster_map left on the device: a zero-copy torch uint8 view [R, n_layers, nh, nw] over the handle's observation
        scratch (valid until the next observation call), ordered after the kernels that fill it.  torch is the container."""
        key = (tuple(layers), int(nw), int(nh))
        cache = self.__dict__.setdefault("_map_views", {})
        ent = cache.get(key)
        if ent is None:
            ent = cache[key] = [np.ascontiguousarray(layers, np.int32), C.c_void_p(), None, None]
        lay, ptr = ent[0], ent[1]
        self._check(self.lib.sg_raster_map_device(self.h, float(width), float(height), key[1], key[2], len(lay), lay.ctypes.data,
                                                  C.byref(ptr)), "sg_raster_map_device")
        self._check(self.lib.sg_synchronize(self.h), "sg_synchronize")
        if ent[2] != ptr.value:  # the scratch moved (first call, or it grew): wrap the new address once
            import torch

            class _Arr:
                __cuda_array_interface__ = dict(shape=(self.R, len(lay), key[2], key[1]), typestr="|u1",
                                                data=(int(ptr.value), False), version=2)

            ent[2], ent[3] = ptr.value, torch.as_tensor(_Arr(), device=f"cuda:{self.cfg.device}")
        return ent[3]

    def tick(self, actions, layers, width=20.0, height=20.0, nw=20, nh=20, torch_out=False):
        """One RL tick as one graph launch (sg_tick): step(1, actions) + terminal_flags + raster_map.  actions [R, 2]: numpy,
        a float64 torch tensor on the device, or None.  Returns (obs [R, n_layers, nh, nw], flags [R]): numpy arrays (bool /
        uint32), or with torch_out zero-copy torch views (uint8 / int32) valid until the next observation call."""
        key = ("tick", tuple(layers), int(nw), int(nh))
        cache = self.__dict__.setdefault("_map_views", {})
        ent = cache.get(key)
        if ent is None:
            ent = cache[key] = [np.ascontiguousarray(layers, np.int32), C.c_void_p(), C.c_void_p(), None, None, None]
        lay, d_obs, d_fl = ent[0], ent[1], ent[2]
        ptr, on_device = None, 0
        if actions is not None and hasattr(actions, "data_ptr") and getattr(actions, "is_cuda", False):
            import torch

            assert actions.dtype == torch.float64 and actions.is_contiguous() and actions.numel() == self.R * 2
            torch.cuda.current_stream(actions.device).synchronize()
            ptr, on_device = actions.data_ptr(), 1
        elif actions is not None:
            actions = np.ascontiguousarray(actions, np.float64).reshape(self.R, 2)
            ptr = actions.ctypes.data
        self._check(self.lib.sg_tick(self.h, ptr, on_device, float(width), float(height), key[2], key[3], len(lay),
                                     lay.ctypes.data, C.byref(d_obs), C.byref(d_fl)), "sg_tick")
        self._check(self.lib.sg_synchronize(self.h), "sg_synchronize")
        shape = (self.R, len(lay), key[3], key[2])
        if not torch_out:
            obs, fl = np.empty(shape, np.uint8), np.empty(self.R, np.uint32)
            self._check(self.lib.sg_copy_to_host(self.h, d_obs, obs.ctypes.data, obs.nbytes), "sg_copy_to_host")
            self._check(self.lib.sg_copy_to_host(self.h, d_fl, fl.ctypes.data, fl.nbytes), "sg_copy_to_host")
            return obs.astype(bool), fl
        if ent[3] != (d_obs.value, d_fl.value):
            import torch

            def view(p, shp, typestr):
                class _Arr:
                    __cuda_array_interface__ = dict(shape=shp, typestr=typestr, data=(int(p), False), version=2)
                return torch.as_tensor(_Arr(), device=f"cuda:{self.cfg.device}")

            ent[3] = (d_obs.value, d_fl.value)
            ent[4], ent[5] = view(d_obs.value, shape, "|u1"), view(d_fl.value, (self.R,), "<i4")
        return ent[4], ent[5]

    def rollout(self, max_steps):
        self._check(self.lib.sg_rollout(self.h, int(max_steps)), "sg_rollout")

    def rollout_async(self, max_steps, do_reset=True):
        self._check(self.lib.sg_rollout_async(self.h, int(max_steps), int(do_reset)), "sg_rollout_async")

    def synchronize(self):
        self._check(self.lib.sg_synchronize(self.h), "sg_synchronize")

    def last_kernel_ms(self):
        ms = C.c_float()
        self._check(self.lib.sg_last_kernel_ms(self.h, C.byref(ms)), "sg_last_kernel_ms")
        return ms.value

    def set_tuning(self, tab_min_steps=-1, chunk_steps=0, overlap=-1):
        """Launch policy (sg_set_tuning); results never depend on it."""
        self._check(self.lib.sg_set_tuning(self.h, int(tab_min_steps), int(chunk_steps), int(overlap)), "sg_set_tuning")

    def set_slicing(self, on=True):
        """sg_set_slicing: may sg_rollout cut the time axis of a small replay-only batch into slices (results never depend on
        it; the intermediate states are not written to memory when it does)."""
        self._check(self.lib.sg_set_slicing(self.h, 2 if on == "always" else int(bool(on))), "sg_set_slicing")

    def last_launch_stats(self):
        """(number of rollout-kernel launches of the last call, the time in ms during which at least one of them ran)."""
        n, ms = C.c_int32(), C.c_float()
        self._check(self.lib.sg_last_launch_stats(self.h, C.byref(n), C.byref(ms)), "sg_last_launch_stats")
        return n.value, ms.value

    def last_launch_gross_ms(self):
        """Sum of the durations of those launches (launches of the two pipelines overlap: more than the time above)."""
        ms = C.c_float()
        self._check(self.lib.sg_last_launch_gross_ms(self.h, C.byref(ms)), "sg_last_launch_gross_ms")
        return ms.value

    def schedule_info(self):
        """The launch schedule of the last rollout / step call (sg_schedule_info): schedule 0 = not the table path, 1 = chunk
        launches, 2 = one persistent launch (csrc/sgym_queue.hpp); its chunks, table-ring buffers and wavefronts; the
        pre-pass wavefronts, blocks and SIMDs; the rollout-kernel launches of the call."""
        v = (C.c_int32 * 8)()
        self._check(self.lib.sg_schedule_info(self.h, v), "sg_schedule_info")
        return dict(schedule=v[0], chunks=v[1], ring=v[2], grid=v[3], ctl_waves=v[4], blocks=v[5], simds=v[6], launches=v[7])

    def last_kernel(self):
        """The rollout entry point the handle launched last (sg_last_kernel), e.g. "sg::rollout_kernel_crowd<4>"."""
        return (self.lib.sg_last_kernel(self.h) or b"").decode()

    def debug_trig32(self, heading):
        """The broad phase's fp32 (sin, cos) of fp64 headings (test hook)."""
        h = np.ascontiguousarray(heading, np.float64).ravel()
        s, c = np.empty(h.size, np.float32), np.empty(h.size, np.float32)
        self._check(self.lib.sg_debug_trig32(self.h, h.size, h.ctypes.data, s.ctypes.data, c.ctypes.data),
                    "sg_debug_trig32")
        return s, c

    # ------------------------------------------------------------------ reads
    def _d2h(self, ptr, shape, dtype):
        out = np.empty(shape, dtype)
        self._check(self.lib.sg_copy_to_host(self.h, ptr, out.ctypes.data, out.nbytes), "sg_copy_to_host")
        return out

    def state(self, raw=False):
        """Host copy of the step-materialised state: dict of [R, E, ...] arrays (NaN = absent).
        `coll` is [R, E] uint64 for up to 64 entities, else [R, E, row_words].
        raw: the stored pose / velocity rows as they are, also for entities that are not in the scene (debugging, tests)."""
        v, R, E, EP = self._view, self.R, self.E, self._view.entity_stride
        n = R * EP
        blocks = self._d2h(v.blocks, (v.n_blocks, v.block_rows, 64), np.float64)

        def rows(f, k=1, dtype=np.float64):
            a = blocks[:, f:f + k, :].view(dtype)                      # [nblk, k, 64]
            return np.moveaxis(a, 1, -1).reshape(-1, k)[:n].reshape(R, EP, k)[:, :E]

        present = rows(L.F_PRESENT, 1, np.uint64)[..., 0] != 0
        scen = self._d2h(v.scen, R, SCEN_DTYPE)
        coll = rows(L.F_COLL, v.row_words, np.uint64)
        return dict(
            poses=rows(L.F_POSE, 6).copy() if raw else np.where(present[..., None], rows(L.F_POSE, 6), np.nan),
            vels=rows(L.F_VEL, 6).copy() if raw else np.where(present[..., None], rows(L.F_VEL, 6), np.nan),
            present=present, dists=rows(L.F_DIST)[..., 0], coll=coll[..., 0] if v.row_words == 1 else coll,
            ctrl_state=rows(L.F_CTRL, 4), force=rows(L.F_FORCE, 2), t=scen["t"].copy(), prev_t=scen["prev_t"].copy(),
            done=scen["done"].astype(bool), n_steps=scen["n_steps"].copy(), noise_pos=scen["noise_pos"].copy(),
        )

    METRIC_DTYPE = np.dtype([("ego_avg_speed", "f8"), ("ego_max_speed", "f8"), ("ego_distance_travelled", "f8"),
                             ("final_t", "f8"), ("n_steps", "i4"), ("done", "i4"), ("n_collisions", "i4"), ("reserved", "i4")])
    EVENT_DTYPE = np.dtype([("t", "f8"), ("scenario", "i4"), ("other", "i4"), ("type", "i4"), ("reserved", "i4")])

    def metrics(self, event_cap=None):
        """(per-scenario metric rows, CollisionMetric events) as structured arrays (sg_metrics / sg_event layout)."""
        cap = self.R * max(self.cfg.event_capacity, 1) if event_cap is None else int(event_cap)
        rows = np.empty(self.R, self.METRIC_DTYPE)   # uninitialised host buffers: the library fills what it reports
        if getattr(self, "_ev_buf", None) is None or len(self._ev_buf) < cap:
            self._ev_buf = np.empty(cap, self.EVENT_DTYPE)   # kept between calls (6 MB for 4096 x 64: not touched unless filled)
        ev = self._ev_buf
        n_ev = C.c_int32()
        self._check(self.lib.sg_read_metrics(self.h, rows.ctypes.data_as(C.POINTER(L.SgMetrics)),
                                             ev.ctypes.data_as(C.POINTER(L.SgEvent)), cap, C.byref(n_ev)), "sg_read_metrics")
        return rows, ev[: n_ev.value].copy()

    def set_rss(self, enabled=True):
        """reset / rollout / step run the RSSDistances callback themselves after the reset and after every step."""
        self._check(self.lib.sg_set_rss(self.h, int(bool(enabled))), "sg_set_rss")

    def rss_update(self, reset=False):
        """RSSDistances.__call__ on the current state of every scenario (after a reset: reset=True)."""
        self._check(self.lib.sg_rss_update(self.h, int(bool(reset))), "sg_rss_update")

    def rss(self):
        """(safe_longitudinal [R] bool, safe_lateral [R] bool, codes [R, E], safe distances [R, E, 2]) of the RSS callback."""
        flags = np.zeros(self.R, np.uint8)
        codes = np.zeros((self.R, self.E), np.int32)
        safe = np.zeros((self.R, self.E, 2))
        self._check(self.lib.sg_rss_read(self.h, flags.ctypes.data, codes.ctypes.data, safe.ctypes.data), "sg_rss_read")
        return (flags & 1) != 0, (flags & 2) != 0, codes, safe

    def collision_points(self, event_cap=None):
        """CollisionPointMetric for the events of metrics(), same order: [n_events, 3] = x, y, relative heading."""
        cap = int(event_cap or self.R * max(int(self.cfg.event_capacity), 1))
        out = np.empty((cap, 3))
        n = C.c_int32()
        self._check(self.lib.sg_read_collision_points(self.h, out.ctypes.data, cap, C.byref(n)), "sg_read_collision_points")
        return out[: n.value].copy()

    def record(self, n_rows):
        """State.recorded_poses for the whole batch: t [n, R], poses [n, R, E, 6]."""
        t = np.empty((n_rows, self.R))
        poses = np.empty((n_rows, self.R, self.E, 6))
        self._check(self.lib.sg_read_record(self.h, int(n_rows), t.ctypes.data, poses.ctypes.data), "sg_read_record")
        return t, poses

    def torch_state(self):
        """Zero-copy torch view [n_blocks, block_rows, 64] (fp64) over the device state blocks;
        field f of entity i is view[i // 64, f, i % 64].  torch is only the container."""
        import torch

        v = self._view

        class _Arr:
            def __init__(self, ptr, shape, typestr):
                self.__cuda_array_interface__ = dict(shape=shape, typestr=typestr, data=(int(ptr), False), version=2)

        return torch.as_tensor(_Arr(v.blocks, (v.n_blocks, v.block_rows, 64), "<f8"), device=f"cuda:{self.cfg.device}")
