import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scenario_gym_amd as sga
import scenario_gym_amd._lib as L
lib = L.load()
cfg = L.SgConfig(0, 256, 16, 0, L.TERM_MAX_LENGTH, 0, 32, 0, 1 / 30)
devs = np.zeros(2, np.int32)
g = C.c_void_p()
print("create", lib.sg_group_create(C.byref(cfg), 2, devs.ctypes.data, C.byref(g)), g, flush=True)
print("size", lib.sg_group_size(g), flush=True)
h = lib.sg_group_handle(g, 0)
print("handle", h, type(h), lib.sg_group_handle.restype, lib.sg_group_handle.argtypes, flush=True)
info = (C.c_int32 * 8)()
print("info rc", lib.sg_schedule_info(h, info), list(info), flush=True)
