"""ctypes binding of libsgym_hip.so (the C ABI of include/sgym.h).

There is no CPU fallback: if the HIP library is missing or cannot be loaded this module raises.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SGYM_LIB") or os.path.join(HERE, "lib", "libsgym_hip.so")  # SGYM_LIB: A/B builds

SG_OK = 0
ABI_VERSION = 5
(KIND_NONE, KIND_REPLAY, KIND_AGENT_REPLAY, KIND_AGENT_PID, KIND_AGENT_VEHICLE, KIND_AGENT_PEDESTRIAN,
 KIND_AGENT_EXTERNAL) = range(7)
TERM_MAX_LENGTH, TERM_COLLISION, TERM_EGO_COLLISION, TERM_EGO_OFF_ROAD = 1, 2, 4, 8
NCTRL = 16
NOISE_OFF, NOISE_STREAM, NOISE_DEVICE = 0, 1, 2  # sg_set_ped_noise
PED_SOCIAL_FORCE, PED_RANDOM_WALK = 0, 1         # sg_set_ped_behaviour
(C_MAX_STEER, C_MAX_ACCEL, C_MAX_SPEED, C_ALLOW_REVERSE, C_STEER_KP, C_STEER_KD, C_ACCEL_KP,
 C_ACCEL_KD, C_ACCEL_KI, C_PED_SPEED_DESIRED, C_PED_MAX_SPEED, C_PED_HEAD_ROT, C_PED_RADIUS) = range(13)

# SG_F_* rows of a state block (include/sgym.h)
F_POSE, F_VEL, F_DIST, F_PRESENT, F_CTRL, F_FORCE, F_COLL = 0, 6, 12, 13, 14, 18, 20  # block_rows = F_COLL + row_words

# every symbol include/sgym.h declares
SYMBOLS = (
    "sg_version", "sg_last_error", "sg_create", "sg_destroy", "sg_upload", "sg_set_social_force", "sg_set_ped_models", "sg_set_ped_behaviour", "sg_set_ped_noise", "sg_reset",
    "sg_set_timestep", "sg_step", "sg_rollout", "sg_rollout_async", "sg_synchronize", "sg_stream",
    "sg_state_view_get", "sg_read_metrics", "sg_read_record", "sg_copy_to_host", "sg_last_kernel_ms",
    "sg_last_launch_stats", "sg_last_launch_gross_ms", "sg_schedule_info", "sg_last_kernel", "sg_debug_trig32", "sg_set_tuning", "sg_set_slicing", "sg_set_external_poses", "sg_future_collision", "sg_raster_entities",
    "sg_set_road_networks", "sg_raster_map", "sg_raster_map_device", "sg_reset_scenarios", "sg_terminal_flags", "sg_tick", "sg_set_collision_tolerance", "sg_read_collision_points", "sg_rss_update", "sg_rss_read", "sg_set_rss",
    "sg_group_create", "sg_group_destroy", "sg_group_size", "sg_group_handle", "sg_group_upload", "sg_group_rollout",
    "sg_group_read_metrics", "sg_group_last_error", "sg_host_alloc", "sg_host_free",
)


class SgConfig(C.Structure):
    _fields_ = [
        ("device", C.c_int32), ("n_scenarios", C.c_int32), ("n_entities", C.c_int32),
        ("persist", C.c_int32), ("terminal_mask", C.c_uint32), ("record_capacity", C.c_int32),
        ("event_capacity", C.c_int32), ("reserved", C.c_int32), ("timestep", C.c_double),
    ]


class SgScenarios(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("kind", "etype", "bbox", "knot_off", "knots", "ctrl", "ego", "t0", "length", "route_off", "routes")]


class SgRoadNetworks(C.Structure):
    _fields_ = [("n_networks", C.c_int32)] + [(n, C.c_void_p) for n in
                                              ("net_of_scenario", "poly_off", "ring_off", "vert_off", "verts", "layers")]


class SgSocialForce(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("relaxation_time", "ped_repulse_V", "ped_repulse_sigma", "ped_attract_C", "sight_weight",
                 "sight_weight_use", "cos_sight", "max_speed_factor", "bias_lon", "bias_lat", "imp_boundary_repulse_U", "imp_boundary_repulse_R")]


class SgPedModel(C.Structure):  # include/sgym.h sg_ped_model
    _fields_ = [("behaviour", C.c_int32), ("reserved", C.c_int32), ("params", SgSocialForce), ("std_lon", C.c_double), ("std_lat", C.c_double)]


class SgStateView(C.Structure):
    _fields_ = [
        ("n_scenarios", C.c_int32), ("n_entities", C.c_int32), ("entity_stride", C.c_int32),
        ("n_blocks", C.c_int32), ("row_words", C.c_int32), ("block_rows", C.c_int32),
        ("blocks", C.c_void_p), ("scen", C.c_void_p),
    ]


class SgMetrics(C.Structure):
    _fields_ = [
        ("ego_avg_speed", C.c_double), ("ego_max_speed", C.c_double),
        ("ego_distance_travelled", C.c_double), ("final_t", C.c_double),
        ("n_steps", C.c_int32), ("done", C.c_int32), ("n_collisions", C.c_int32), ("reserved", C.c_int32),
    ]


class SgEvent(C.Structure):
    _fields_ = [("t", C.c_double), ("scenario", C.c_int32), ("other", C.c_int32),
                ("type", C.c_int32), ("reserved", C.c_int32)]


_lib = None


def source_sha16() -> str:
    """sha256[:16] of the kernel sources the library is built from (csrc/*.hpp, *.hip, include/sgym.h).  Profiles committed
    under profiles/ carry it; bench.py reports a profile's numbers only when it matches the tree it runs from."""
    import hashlib

    h = hashlib.sha256()
    csrc = os.path.join(HERE, "csrc")
    units = sorted(f for f in os.listdir(csrc) if f.endswith((".hpp", ".hip")))  # one object per kernel family (csrc/Makefile)
    for rel in [os.path.join("csrc", f) for f in units] + ["../include/sgym.h"]:
        with open(os.path.join(HERE, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pinned_empty(shape, dtype=np.float64, device=0):
    """An uninitialised numpy array in page-locked host memory (sg_host_alloc): the copy engine reads it directly at the PCIe
    rate, where ordinary memory goes through the runtime's staging copies.  Freed when the array (and its views) are gone."""
    import weakref

    lib = load()
    dtype = np.dtype(dtype)
    n = int(np.prod(shape))
    ptr = C.c_void_p()
    rc = lib.sg_host_alloc(int(device), max(n * dtype.itemsize, 1), C.byref(ptr))
    if rc != 0:
        raise RuntimeError(f"sg_host_alloc({n * dtype.itemsize} bytes) failed: {rc}")
    buf = (C.c_char * max(n * dtype.itemsize, 1)).from_address(ptr.value)
    weakref.finalize(buf, lib.sg_host_free, ptr.value)  # the array below keeps `buf` alive through .base
    return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)


def load():
    """Load libsgym_hip.so and declare its prototypes.  Raises if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C scenario_gym_amd/csrc` "
            "(or __graft_entry__.build()).  scenario_gym_amd has no CPU fallback."
        )
    try:  # PyTorch bundles its own libamdhip64: let it load first so that both share ONE HIP runtime
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    H = C.c_void_p
    lib.sg_version.restype = C.c_int
    lib.sg_last_kernel.restype = C.c_char_p
    lib.sg_last_kernel.argtypes = [H]
    lib.sg_last_error.restype = C.c_char_p
    lib.sg_last_error.argtypes = [H]
    lib.sg_create.argtypes = [C.POINTER(SgConfig), C.POINTER(H)]
    lib.sg_destroy.argtypes = [H]
    lib.sg_upload.argtypes = [H, C.POINTER(SgScenarios)]
    lib.sg_set_social_force.argtypes = [H, C.POINTER(SgSocialForce)]
    lib.sg_set_ped_models.argtypes = [H, C.c_int32, C.POINTER(SgPedModel), C.c_void_p]
    lib.sg_set_ped_behaviour.argtypes = [H, C.c_int32]
    lib.sg_set_ped_noise.argtypes = [H, C.c_int32, C.c_double, C.c_double, C.c_void_p, C.c_int64, C.c_uint64]
    lib.sg_reset.argtypes = [H]
    lib.sg_set_timestep.argtypes = [H, C.c_double]
    lib.sg_step.argtypes = [H, C.c_int32, C.c_void_p, C.c_int32]
    lib.sg_rollout.argtypes = [H, C.c_int32]
    lib.sg_rollout_async.argtypes = [H, C.c_int32, C.c_int32]
    lib.sg_synchronize.argtypes = [H]
    lib.sg_stream.restype = C.c_void_p
    lib.sg_stream.argtypes = [H]
    lib.sg_state_view_get.argtypes = [H, C.POINTER(SgStateView)]
    lib.sg_read_metrics.argtypes = [H, C.POINTER(SgMetrics), C.POINTER(SgEvent), C.c_int32, C.POINTER(C.c_int32)]
    lib.sg_read_record.argtypes = [H, C.c_int32, C.c_void_p, C.c_void_p]
    lib.sg_copy_to_host.argtypes = [H, C.c_void_p, C.c_void_p, C.c_uint64]
    lib.sg_last_kernel_ms.argtypes = [H, C.POINTER(C.c_float)]
    lib.sg_last_launch_stats.argtypes = [H, C.POINTER(C.c_int32), C.POINTER(C.c_float)]
    lib.sg_last_launch_gross_ms.argtypes = [H, C.POINTER(C.c_float)]
    lib.sg_schedule_info.argtypes = [H, C.POINTER(C.c_int32)]
    lib.sg_set_tuning.argtypes = [H, C.c_int32, C.c_int32, C.c_int32]
    lib.sg_set_slicing.argtypes = [H, C.c_int32]
    lib.sg_host_alloc.argtypes = [C.c_int32, C.c_uint64, C.POINTER(C.c_void_p)]
    lib.sg_host_free.argtypes = [C.c_void_p]
    lib.sg_set_external_poses.argtypes = [H, C.c_void_p]
    lib.sg_future_collision.argtypes = [H, C.c_double, C.c_int32, C.c_void_p]
    lib.sg_raster_entities.argtypes = [H, C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_void_p]
    lib.sg_set_road_networks.argtypes = [H, C.POINTER(SgRoadNetworks)]
    lib.sg_reset_scenarios.argtypes = [H, C.c_void_p]
    lib.sg_set_collision_tolerance.argtypes = [H, C.c_double]
    lib.sg_rss_update.argtypes = [H, C.c_int32]
    lib.sg_set_rss.argtypes = [H, C.c_int32]
    lib.sg_rss_read.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.sg_read_collision_points.argtypes = [H, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    lib.sg_group_create.argtypes = [C.POINTER(SgConfig), C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.sg_group_destroy.argtypes = [C.c_void_p]
    lib.sg_group_size.argtypes = [C.c_void_p]
    lib.sg_group_handle.argtypes = [C.c_void_p, C.c_int32]
    lib.sg_group_handle.restype = C.c_void_p
    lib.sg_group_upload.argtypes = [C.c_void_p, C.POINTER(SgScenarios)]
    lib.sg_group_rollout.argtypes = [C.c_void_p, C.c_int32]
    lib.sg_group_read_metrics.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    lib.sg_group_last_error.argtypes = [C.c_void_p]
    lib.sg_group_last_error.restype = C.c_char_p
    lib.sg_tick.argtypes = [H, C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                            C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    lib.sg_terminal_flags.argtypes = [H, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.sg_raster_map.argtypes = [H, C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.sg_raster_map_device.argtypes = [H, C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.sg_debug_trig32.argtypes = [H, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    for name in SYMBOLS:
        if name not in ("sg_last_error", "sg_last_kernel", "sg_stream", "sg_version", "sg_group_handle", "sg_group_last_error"):  # (pointers / strings)
            getattr(lib, name).restype = C.c_int
    if lib.sg_version() != ABI_VERSION:
        raise RuntimeError(f"libsgym_hip.so ABI {lib.sg_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib
