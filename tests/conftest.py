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
