"""scenario_gym_amd: MI355X-native batched rollout engine behind scenario_gym's Python API."""
from .engine import PackedScenarios, RolloutEngine  # noqa: F401

__version__ = "0.1.0"
