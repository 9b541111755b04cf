import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def bits_equal(a, b):
    """Bitwise equality of two float64 arrays (NaN pattern must agree too; -0.0 == +0.0)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if a.shape != b.shape:
        return False
    na, nb = np.isnan(a), np.isnan(b)
    return bool(np.array_equal(na, nb) and np.array_equal(a[~na], b[~nb]))


def scenario_arrays(g, prefix):
    """Numeric scenario content stored by make_golden.export_scenario."""
    knots = g[prefix + "/knots"]
    off = g[prefix + "/knot_off"]
    ego = int(g[prefix + "/ego"])
    return dict(
        knot_off=off,
        knots=knots,
        bbox=g[prefix + "/bbox"],
        etype=g[prefix + "/etype"],
        ego=ego,
        length=float(g[prefix + "/length"]),
        t0=max(0.0, float(knots[off[ego], 0])),  # ScenarioGym.get_start_time, scenario_gym.py:213-215
    )


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.build()
    return O


@pytest.fixture(autouse=True)
def _scrambled_device_memory(request):
    """SG_SCRAMBLE=<GiB>: before every GPU test, that much free device memory is filled with a pattern and released, so that
    what the library allocates next is NOT what a previous run of the same suite left at the same addresses (the first suite run
    on a fresh box sees other people's bytes there; every later run sees its own, valid-looking ones -- a read of memory nobody
    wrote only shows in the first).  Patterns rotate: NaNs, random bytes, huge integers."""
    gib = float(os.environ.get("SG_SCRAMBLE", "0") or 0)
    if gib > 0 and request.node.get_closest_marker("gpu") is not None:
        import torch

        n = int(gib * (1 << 30))
        k = getattr(_scrambled_device_memory, "k", 0)
        _scrambled_device_memory.k = k + 1
        parts = []
        for i in range(8):  # (several pieces: the allocator hands the library blocks out of any of them)
            t = torch.empty(n // 8, dtype=torch.uint8, device="cuda:0")
            if k % 3 == 0:
                t.fill_(0xFF)
            elif k % 3 == 1:
                t.random_(0, 256)
            else:
                t.fill_(0x7F)
            parts.append(t)
        torch.cuda.synchronize()
        del parts, t
        torch.cuda.empty_cache()
    yield
