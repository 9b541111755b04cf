#!/usr/bin/env python3
"""Golden vectors for scenario actions on the per-tick path (State.update_actions / apply_action / entity_state /
action_apply_times: scenario_gym/state/state.py:150-160, 241-266; scenario/actions.py:12-168), from the REAL reference (build
container only):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden_actions.py   ->  tests/golden/actions.npz

The shipped scenario that carries OpenSCENARIO UserDefinedActions (1518e754...) gets UpdateStateVariableActions added --
before its start, exactly ON step times, between steps, after its end -- and is stepped through the reference's ScenarioGym
at two timesteps: per action the time State.action_apply_times records (NaN: never applied), per entity the final
State.entity_state as JSON, the per-step get_entity_data(ego)[-1], and Scenario.translate's effect on the action times.
Only DATA is written."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path[:0] = [os.path.join(HERE, "_refstubs"), "/root/reference"]

import numpy as np  # noqa: E402

import scenario_gym  # noqa: E402
from scenario_gym import ScenarioGym  # noqa: E402
from scenario_gym.scenario.actions import UpdateStateVariableAction  # noqa: E402
from scenario_gym.xosc_interface import import_scenario  # noqa: E402

assert scenario_gym.__version__ == "0.3.1"
NAME = "1518e754-318f-4847-8a30-2dce552b4504"
PATH = f"/root/reference/tests/input_files/Scenarios/{NAME}.xosc"


def added_actions(s, dt):
    t0 = max(0.0, s.ego.trajectory.min_t)
    refs = [e.ref for e in s.entities]
    k3 = t0
    for _ in range(3):
        k3 = k3 + dt  # State.t after three steps: an action exactly ON a step time (">" must wait one more step)
    return [
        UpdateStateVariableAction(t0 - 1.0, "UpdateStateVariableAction", refs[0], {"mode": "early", "n": 1}),
        UpdateStateVariableAction(k3, "UpdateStateVariableAction", refs[0], {"mode": "on_step", "gear": 3}),
        UpdateStateVariableAction(t0 + 2.5 * dt, "UpdateStateVariableAction", refs[-1], {"lights": True}),
        UpdateStateVariableAction(t0 + 7.25 * dt, "UpdateStateVariableAction", refs[0], {"mode": "late"}),
        UpdateStateVariableAction(t0 + 4.0, "UpdateStateVariableAction", "nobody", {"ghost": 1}),
        UpdateStateVariableAction(s.length + 100.0, "UpdateStateVariableAction", refs[0], {"never": 1}),
    ]


def main():
    out = {"name": np.array(NAME)}
    for tag, dt in (("dt30", 1.0 / 30.0), ("dt10", 0.1)):
        s = import_scenario(PATH)
        for a in added_actions(s, dt):
            s.add_action(a, inplace=True)
        acts = list(s.actions)
        out[f"{tag}/added"] = np.array(json.dumps([a.to_dict() for a in added_actions(s, dt)]))
        out[f"{tag}/classes"] = np.array([type(a).__name__ for a in acts])
        out[f"{tag}/t"] = np.array([a.t for a in acts])
        out[f"{tag}/entity_ref"] = np.array([a.entity_ref for a in acts])
        import warnings

        gym = ScenarioGym(timestep=dt)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # (the action for "nobody" warns when it is applied)
            gym.set_scenario(s)
            per_step, times = [], []
            per_step.append(json.dumps(gym.state.get_entity_data(s.ego)[-1], sort_keys=True))
            times.append(gym.state.t)
            after_reset = [gym.state.action_apply_times[a] for a in acts]
            while not gym.state.is_done:
                gym.step()
                per_step.append(json.dumps(gym.state.get_entity_data(s.ego)[-1], sort_keys=True))
                times.append(gym.state.t)
        out[f"{tag}/apply_times_after_reset"] = np.array(after_reset, np.float64)
        out[f"{tag}/apply_times"] = np.array([gym.state.action_apply_times[a] for a in acts], np.float64)
        out[f"{tag}/entity_state"] = np.array(json.dumps({e.ref: gym.state.entity_state[e] for e in s.entities}, sort_keys=True))
        out[f"{tag}/ego_state_per_step"] = np.array(per_step)
        out[f"{tag}/clock"] = np.array(times, np.float64)
        out[f"{tag}/n_unapplied"] = np.int64(len(gym.state.unapplied_actions))
        shifted = s.reset_start()
        out[f"{tag}/t_after_reset_start"] = np.array([a.t for a in shifted.actions])
        out[f"{tag}/to_dict_actions"] = np.array(json.dumps([a.to_dict() for a in s.actions]))
        print(tag, len(acts), "actions,", int(np.isfinite(out[f"{tag}/apply_times"]).sum()), "applied,", len(times) - 1, "steps",
              out[f"{tag}/entity_state"])
    np.savez_compressed(os.path.join(os.environ.get("SG_GOLDEN_OUT", HERE), "actions.npz"), **out)


if __name__ == "__main__":
    main()
