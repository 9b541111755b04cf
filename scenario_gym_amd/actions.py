"""Scenario actions on the host side of the per-tick path (reference scenario/actions.py:12-168, state/state.py:241-266).

State.step ends with update_actions(): every action of the scenario that has not been applied yet and whose
trigger_condition(state) holds is applied to the state (apply_action -> action.apply(state, entity)) and its application time
recorded in State.action_apply_times.  None of the reference's action classes touches a pose -- UserDefinedAction does
nothing, UpdateStateVariableAction writes State.entity_state[entity] -- so the device rollout is unaffected and the actions run
in Python after every tick (gym.step) or, for the time-triggered classes below, from the clock of a whole device rollout
(BatchedScenarioGym.rollout: the same application times and entity states, bit for bit, as stepping one by one)."""
from copy import deepcopy
from typing import Any, Dict, Optional


class ScenarioAction:
    """Base class (scenario/actions.py:12-78): subclasses give `_apply(state, entity)` and `trigger_condition(state)`."""

    def __init__(self, action_class: str, entity_ref: str, action_variables: Dict[str, Any]):
        self.action_class = action_class
        self.entity_ref = entity_ref
        self.action_variables = action_variables

    def apply(self, state, entity) -> None:
        self._apply(state, entity)

    def _apply(self, state, entity) -> None:
        raise NotImplementedError

    def trigger_condition(self, state) -> bool:
        raise NotImplementedError

    def copy(self):
        return deepcopy(self)

    def translate(self, x, inplace: bool = False):
        return self if inplace else self.copy()

    def to_dict(self) -> Dict[str, Any]:
        return {"action_class": self.action_class, "entity_ref": self.entity_ref, "action_variables": self.action_variables}

    @classmethod
    def from_dict(cls, data: Dict[str, Any]):
        return cls(data["action_class"], data["entity_ref"], data["action_variables"])


class FixedTAction(ScenarioAction):
    """Applied at the first step whose time has reached `t` (scenario/actions.py:81-126)."""

    strict = False  # trigger: state.t >= t

    def __init__(self, t: float, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.t = t

    def trigger_condition(self, state) -> bool:
        return state.t > self.t if self.strict else state.t >= self.t

    def translate(self, x, inplace: bool = False):
        act = self if inplace else self.copy()
        act.t += x[0]
        return act

    def to_dict(self) -> Dict[str, Any]:
        data = super().to_dict()
        data["t"] = self.t
        return data

    @classmethod
    def from_dict(cls, data: Dict[str, Any]):
        return cls(data["t"], data["action_class"], data["entity_ref"], data["action_variables"])


class UserDefinedAction(FixedTAction):
    """What an OpenSCENARIO UserDefinedAction becomes (xosc_interface/read.py:224-245); applying it does nothing (:129-134)."""

    def _apply(self, state, entity) -> None:
        pass


class UpdateStateVariableAction(FixedTAction):
    """Sets State.entity_state[entity][key] = value for its action variables (:137-168); triggers when state.t > t."""

    strict = True

    def _apply(self, state, entity) -> None:
        if entity is not None:
            if state.entity_state[entity] is None:
                state.entity_state[entity] = {}
            for k, v in self.action_variables.items():
                state.entity_state[entity][k] = v

    def to_dict(self) -> Dict[str, Any]:
        return {"t": self.t, "action_class": self.action_class, "entity_ref": self.entity_ref,
                "action_variables": self.action_variables}


def ScenarioActionRecord(t: float, action_class: str, entity_ref: str, action_variables: Dict[str, Any],
                         kind: str = "UpdateStateVariableAction"):
    """(earlier name of the two concrete classes; kept for callers that built records by kind)"""
    cls = UserDefinedAction if kind == "UserDefinedAction" else UpdateStateVariableAction
    return cls(t, action_class, entity_ref, action_variables)


def time_triggered(action) -> Optional[bool]:
    """True / False: the action is one of the classes above WITHOUT an overridden trigger or apply, i.e. its application
    step follows from the clock alone (strict or not); None: anything else (evaluated tick by tick)."""
    cls = type(action)
    if cls in (UserDefinedAction, UpdateStateVariableAction, FixedTAction):
        return bool(cls.strict)
    return None
