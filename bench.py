#!/usr/bin/env python3
"""Batched-rollout benchmark: entity-steps/s on BASELINE.json's 4096 scenarios x 64 entities x 10k steps.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One bench "step" = one full pass of the hot path over the batch: sg_rollout of R x E x T
(reset + T simulated steps, every scenario running to its end) with the scenarios resident in HBM.
Every rank owns its own R scenarios (weak scaling: replicas are independent); RCCL is
used only to dispatch the run configuration and to collect the per-replica metric rows.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ALG = 114.0        # algorithmic bytes per entity-step: pose 48 + vel 48 + dist 8 + row 8 + knots 2
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(workload, seconds_budget=20.0):
    """The CPU oracle (a port of the reference's algorithm, NOT the thing measured) on a bounded
    sample of the same workload: one scenario per host thread, the same E, shortened T."""
    import concurrent.futures as cf

    import numpy as np

    import scenario_gym_amd._lib as L
    from oracle import oracle as O
    from scenario_gym_amd import synthetic
    from scenario_gym_amd.packing import unpack_scenario

    O.build()
    O.lib()
    cores = os.cpu_count() or 1
    E, T = workload["E"], workload["T"]
    if workload.get("crowd"):
        t_sample, n_scen = min(T, 1000), max(cores, 1)
        packed = synthetic.make_crowd(n_scen, E, n_steps=t_sample)
    else:
        t_sample, n_scen = T, max(cores, 1) * 4
        packed = synthetic.make_batch(n_scen, E, n_steps=t_sample, ego_kind=workload["ego_kind"])
    scen = [unpack_scenario(packed, r) for r in range(n_scen)]

    def one(s):
        o = O.rollout(s["knot_off"], s["knots"], s["bbox"], s["etype"], s["kind"], s["ego"], s["t0"],
                      s["length"], workload["dt"], ctrl=s["ctrl"], max_steps=t_sample, record=False,
                      route_off=s.get("route_off"), routes=s.get("routes"))
        return o["n_steps"]

    one(scen[0])  # warm
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(cores) as ex:  # ctypes releases the GIL: real parallelism
        steps = list(ex.map(one, scen))
    dt = time.perf_counter() - t0
    return {
        "value": float(sum(steps)) * E / dt, "unit": "entity-steps/s", "cores": cores, "kind": "port",
        "sample": f"{n_scen} scenarios x {E} entities x {t_sample} steps of the same seeded family, "
                  f"{cores} threads, C oracle (oracle/sgym_oracle.c), {dt:.1f}s",
    }


def kernel_name(E, crowd, controlled):
    """Entry point the library launches for this shape (sgym_hip.hip launch_variant): tile lanes G, wavefronts per tile;
    `controlled`: the batch has PID / vehicle agents (their pre-pass table is replayed by rollout_kernel_tab)."""
    G, WV = min(64, max(4, 1 << (E - 1).bit_length())), (1 if E <= 64 else 2 if E <= 128 else 4)
    if crowd:
        return f"sg::rollout_kernel<{max(G, 16) if WV == 1 else G}, {WV}, true, false>"
    return f"sg::rollout_kernel_tab<{G}>" if (WV == 1 and controlled) else f"sg::rollout_kernel<{G}, {WV}, false, true>"


def measured_traffic(R, E, T, launches_per_rollout=1.0):
    """HBM bytes per rollout-kernel launch from the PMC passes committed under profiles/ (FETCH_SIZE +
    WRITE_SIZE, MI355X_MICROARCH.md HBM section), if that profile was taken on this workload."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        rec = json.load(f)
    if (rec.get("scenarios"), rec.get("entities"), rec.get("sim_steps")) != (R, E, T):
        return None
    if "hbm_bytes_per_rollout" in rec:  # summed over the launches of one rollout
        return rec["hbm_bytes_per_rollout"] / launches_per_rollout
    return rec.get("hbm_bytes_per_launch")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scenarios", type=int, default=None, help="scenarios per GPU (default: the workload's own, 4096 for c3)")
    ap.add_argument("--entities", type=int, default=None)
    ap.add_argument("--sim-steps", type=int, default=10000)
    ap.add_argument("--ego", default="pid", choices=["pid", "replay"])
    ap.add_argument("--workload", default="c3", choices=["c3", "c2", "c5"],
                    help="BASELINE.json configs: c3 = 4096x64 PID ego (default, the headline), c2 = 256x16 replay, "
                         "c5 = 1024x256 social-force crowd")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --scenarios per GPU; strong: --scenarios in total, split evenly over the ranks "
                         "(BASELINE.json configs[3] read literally: 4096 replicas over 8 GPUs = 512 per GPU, half a "
                         "wavefront per SIMD)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch

    import scenario_gym_amd as sga
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import synthetic

    from scenario_gym_amd import distributed as D

    shape = {"c2": (256, 16), "c3": (4096, 64), "c5": (1024, 256)}[args.workload]  # BASELINE.json configs
    args.scenarios = args.scenarios or shape[0]  # (explicit --scenarios / --entities: size sweeps of the same family)
    args.entities = args.entities or shape[1]
    if args.workload == "c2":
        args.ego = "replay"
    rank, world, local_rank, dist = D.init()
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    dev = torch.device("cuda", local_rank)
    if args.scaling == "strong":
        if args.scenarios % world:
            raise SystemExit(f"--scaling strong: {args.scenarios} scenarios do not split evenly over {world} ranks")
        args.scenarios //= world

    dt = 1.0 / 30.0
    ego_kind = L.KIND_AGENT_PID if args.ego == "pid" else L.KIND_AGENT_REPLAY
    # dispatch: rank 0 broadcasts the run configuration (RCCL); each rank generates exactly its own shard
    R, E, T, ego_kind, seed = D.dispatch_config(
        [args.scenarios, args.entities, args.sim_steps, ego_kind, synthetic.SEED], dist)

    crowd = args.workload == "c5"
    if crowd:
        packed = synthetic.make_crowd(R, E, n_steps=T, timestep=dt, seed=seed, first_scenario=rank * R)
    else:
        packed = synthetic.make_batch(R, E, n_steps=T, timestep=dt, ego_kind=ego_kind, seed=seed,
                                      first_scenario=rank * R)
    eng = sga.RolloutEngine(R, E, timestep=dt, terminal_conditions=["max_length"], event_capacity=64,
                            device=local_rank)
    eng.upload(packed)
    del packed

    def one_pass():
        eng.rollout_async(T, do_reset=True)
        eng.synchronize()
        rows, _ = eng.metrics()
        # collection: every rank's per-replica metric rows to rank 0 (RCCL gather)
        D.gather_rows(np.stack([rows["ego_avg_speed"], rows["ego_max_speed"], rows["ego_distance_travelled"],
                                rows["n_collisions"].astype(np.float64), rows["n_steps"].astype(np.float64)],
                               axis=1), dist)
        n_launch, launch_ms = eng.last_launch_stats()
        return rows, (eng.last_kernel_ms(), n_launch, launch_ms)

    for _ in range(args.warmup):
        one_pass()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ms = []
    ent_steps = 0
    for _ in range(args.steps):
        rows, ms = one_pass()
        kernel_ms.append(ms)
        ent_steps += int(rows["n_steps"].sum()) * E
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    elapsed = D.max_over_ranks(elapsed, dist)
    total = D.sum_over_ranks(float(ent_steps), dist)

    if rank == 0:
        # dominant kernel = sg::rollout_kernel.  A long rollout is cut into chunks of steps (one launch each, so that the
        # controller pre-pass of the next chunk overlaps it): per-launch units and duration are averages over the
        # launches of the timed passes, each launch timed with its own HIP event pair on the handle's stream.
        n_launches = sum(k[1] for k in kernel_ms)
        per_launch = ent_steps / n_launches
        avg_ms = sum(k[2] for k in kernel_ms) / n_launches
        rollout_ms = sum(k[0] for k in kernel_ms) / len(kernel_ms)  # everything one sg_rollout enqueues
        b_alg = 154.0 if crowd else B_ALG  # +24 B of collision row words, +16 B force vector
        achieved = per_launch * b_alg / (avg_ms * 1e-3) / 1e9
        line = {
            "metric": "entity-steps/sec (batched rollout)",
            "value": total / elapsed,
            "unit": "entity-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (f"{R} scenarios x {E} pedestrians x {T} steps per GPU, PedestrianAgent + SocialForce "
                             "(radius 3 m, noise off) + PedestrianController, all-pairs OBB collisions, CollisionMetric, "
                             "terminal max_length (BASELINE.json configs[4])") if crowd else
                            (f"{R} scenarios x {E} entities x {T} steps per GPU, "
                             f"{'PIDAgent' if ego_kind == L.KIND_AGENT_PID else 'ReplayTrajectoryAgent'} ego + batch replay "
                             "others, all-pairs OBB collisions, CollisionMetric + EgoAvgSpeed/MaxSpeed/DistanceTravelled, "
                             f"terminal max_length (BASELINE.json configs[{1 if args.workload == 'c2' else 2}])"),
                "scenarios_per_gpu": R, "entities": E, "sim_steps": T, "timestep": dt,
                "sharding": f"replicas x{world}, no data-path collective",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(R, E, T, n_launches / args.steps),
                "kernel": kernel_name(E, crowd, ego_kind == L.KIND_AGENT_PID),
                "kernel_ms": avg_ms, "launches_per_rollout": n_launches / args.steps, "rollout_device_ms": rollout_ms,
                "bytes_per_entity_step": b_alg, "entity_steps_per_launch": per_launch,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(dict(E=E, T=T, dt=dt, ego_kind=ego_kind, crowd=crowd))
        print(json.dumps(line))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
