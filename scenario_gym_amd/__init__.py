"""scenario_gym_amd: MI355X-native batched rollout engine behind scenario_gym's Python API."""
from .agent import (  # noqa: F401
    Action,
    Agent,
    CombinedSensor,
    Controller,
    EgoLocalizationSensor,
    ExternalVehicleAgent,
    FutureCollisionDetector,
    GlobalCollisionDetector,
    PIDController,
    RasterizedMapSensor,
    ReplayTrajectoryController,
    Sensor,
    VehicleController,
    PedestrianAgent,
    PedestrianBehaviour,
    PIDAgent,
    RandomWalk,
    RandomWalkParameters,
    ReplayTrajectoryAgent,
    SocialForce,
    SocialForceParameters,
    TeleportAction,
    VehicleAction,
)
from .engine import PackedScenarios, RolloutEngine  # noqa: F401
from .observation import (  # noqa: F401
    CollisionObservation, FutureCollisionObservation, MapObservation, Observation, SingleEntityObservation, combine_observations,
)
from .entity import BoundingBox, CatalogEntry, Entity, MiscObject, Pedestrian, Vehicle  # noqa: F401
from .gym import BatchedScenarioGym, ScenarioGym  # noqa: F401
from .metrics import (  # noqa: F401
    RSS, CollisionMetric, CollisionPointMetric, EgoAvgSpeed, EgoDistanceTravelled, EgoMaxSpeed, Metric, RSSDistances, StateCallback,
)
from .actions import FixedTAction, ScenarioAction, UpdateStateVariableAction, UserDefinedAction  # noqa: F401
from .scenario import Scenario  # noqa: F401
from .state import TERMINAL_CONDITIONS, State  # noqa: F401
from .trajectory import Trajectory  # noqa: F401
from .road_network import RoadNetwork  # noqa: F401
from .route import RouteFinder  # noqa: F401
from .vector_env import VectorScenarioEnv  # noqa: F401

__version__ = "0.1.0"
