#!/usr/bin/env python3
"""Algorithmic fp64 flops per entity-step of the bench workloads, by category -- the numerator of bench.py's vector-ALU
roofline (SURVEY.md 8d "Algorithmic flops").

    python tools/count_flops.py [--workloads c3 c5 c2] [--scenarios K]

Runs K scenarios of the EXACT bench batch (same generator, same seed, full horizon) through the counter build of the CPU
oracle (oracle/sgym_oracle.c with -DSGO_COUNT_FLOPS: per-site constants of the minimal formulation x the calls the run makes;
the convention is at the top of that file) and writes profiles/flops_<workload>.json.  CPU only, no GPU, no reference.
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SGYM_ORACLE_LIB"] = os.path.join(ROOT, "oracle", "_build", "libsgym_oracle_count.so")
sys.path.insert(0, ROOT)

CATS = ["lerp", "statistics", "sincos", "corners", "pair_search", "sat", "controller", "metrics", "ped_goal", "ped_entity",
        "ped_pair", "ped_move", "entity_steps"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", nargs="+", default=["c3", "c5", "c2"])
    ap.add_argument("--scenarios", type=int, default=None)
    ap.add_argument("--sim-steps", type=int, default=10000)
    a = ap.parse_args()
    import numpy as np

    import scenario_gym_amd._lib as L
    from oracle import oracle as O
    from scenario_gym_amd import synthetic
    from scenario_gym_amd.packing import unpack_scenario

    O.build(force=not os.path.exists(os.environ["SGYM_ORACLE_LIB"]))
    lib = O.lib()
    lib.sgo_flops_read.restype = ctypes.c_int
    out = (ctypes.c_ulonglong * len(CATS))()
    dt, T = 1.0 / 30.0, a.sim_steps
    shapes = {"c3": (4096, 64, L.KIND_AGENT_PID), "c2": (256, 16, L.KIND_AGENT_REPLAY), "c5": (1024, 256, None)}
    for wl in a.workloads:
        R, E, ego = shapes[wl]
        K = a.scenarios or (8 if wl == "c5" else 64)
        # scenarios spread over the batch: the generator works in chunks of synthetic.CHUNK scenarios
        picks = sorted({int(i) for i in np.linspace(0, R - 1, K)})
        lib.sgo_flops_read(out, 1)
        steps = 0
        for r in picks:
            first = (r // synthetic.CHUNK) * synthetic.CHUNK
            if wl == "c5":
                packed = synthetic.make_crowd(synthetic.CHUNK, E, n_steps=T, timestep=dt, first_scenario=first)
            else:
                packed = synthetic.make_batch(synthetic.CHUNK, E, n_steps=T, timestep=dt, ego_kind=ego, first_scenario=first)
            s = unpack_scenario(packed, r - first)
            o = O.rollout(s["knot_off"], s["knots"], s["bbox"], s["etype"], s["kind"], s["ego"], s["t0"], s["length"], dt,
                          ctrl=s["ctrl"], max_steps=T, route_off=s.get("route_off"), routes=s.get("routes"))
            steps += o["n_steps"]
        lib.sgo_flops_read(out, 1)
        cnt = dict(zip(CATS, [int(v) for v in out]))
        es = cnt.pop("entity_steps")
        assert es == steps * E, (es, steps, E)
        per = {k: v / es for k, v in cnt.items()}
        total = sum(per.values())
        rec = {
            "workload": wl, "scenarios_of_batch": R, "entities": E, "sim_steps": T, "timestep": dt, "seed": synthetic.SEED,
            "sampled_scenarios": picks, "entity_steps_counted": es,
            "flops_per_entity_step": per, "flops_per_entity_step_total": total,
            "flops_per_entity_step_without_pair_search": total - per["pair_search"],
            "convention": "add / sub / mul / div / sqrt / rint / deciding compare = 1, fma = 2, index work 0; minimal formulation "
                          "(prepared slopes, per-entity terms once per entity-step); pair_search = 6 flops per unordered pair of "
                          "present entities (SURVEY 8d) [+ 1 per unordered pedestrian pair for the neighbour radius]; sat as executed "
                          "with early exits on the pairs whose axis-aligned boxes overlap, one evaluation per unordered pair; "
                          "oracle/sgym_oracle.c (-DSGO_COUNT_FLOPS), tools/count_flops.py",
        }
        path = os.path.join(ROOT, "profiles", f"flops_{wl}.json")
        with open(path, "w") as f:
            json.dump(rec, f, indent=1)
        print(wl, f"{total:.1f} flops / entity-step ({total - per['pair_search']:.1f} without the pair search)",
              {k: round(v, 2) for k, v in per.items()})


if __name__ == "__main__":
    main()
