#!/usr/bin/env python3
"""Batched-rollout benchmark: entity-steps/s on BASELINE.json's 4096 scenarios x 64 entities x 10k steps.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One bench "step" = one full pass of the hot path over the batch: sg_rollout of R x E x T
(reset + T simulated steps, every scenario running to its end) with the scenarios resident in HBM.
Every rank owns its own R scenarios (weak scaling: replicas are independent); RCCL is
used only to dispatch the run configuration and to collect the per-replica metric rows.  With N > 1 the
same run also times BASELINE.json configs[3] read literally (the 4096 scenarios split over the N ranks:
"strong") and reports it beside the headline.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (The table path is ONE persistent launch on ONE stream since round 5 -- csrc/sgym_queue.hpp: the line no longer depends on
# how many hardware queues HIP gives the process, and nothing here sets GPU_MAX_HW_QUEUES any more.)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# fp64 vector ALU: 256 CUs x 4 SIMDs x 16 lanes x 2 (fma) x 2.4 GHz (AMD's datasheet figure for MI355X; the
# live figure of tools/valu_peak.hip is reported beside it)
FP64_VALU_PEAK_TFLOPS = 78.6
# whose counted algorithmic flops (profiles/flops_<key>.json, tools/count_flops.py) a workload is priced with
FLOPS_OF = {"c3": "c3", "c3s": "c3", "c2": "c2", "c2s": "c2", "c5": "c5", "c5mix": "c5", "c5roads": "c5", "c3rss": "c3"}
# BASELINE.json configs: scenarios, entities; algorithmic bytes per entity-step (pose 48 + velocity 48 +
# distance 8 + collision row 8 x words + knots 2 [+ force 16]); bytes the kernel actually stores per steady entity-step
# (unchanged z / pitch / roll rows are not stored again: x, y, h of pose and velocity, distance,
# presence, the collision row words [+ the force])
WORKLOADS = {
    "c2": dict(R=256, E=16, b_alg=114.0, stored=72.0, config=1),
    # the same batch through the time-sliced replay path (sg_set_slicing): final state + metrics + events, NO per-step state in
    # memory -- its own accounting: per entity-step one 8-byte |delta pose| term written by the slices and read by the ordered pass
    "c2s": dict(R=256, E=16, b_alg=16.0, stored=8.0, config=1, sliced=True),
    "c3": dict(R=4096, E=64, b_alg=114.0, stored=72.0, config=2),
    # one GPU's shard of BASELINE.json configs[3] read literally (4096 scenarios over 8 GPUs = 512 each), PID egos, through the
    # time-sliced path: the controller pre-pass fills one table for the whole horizon, the slices replay it.  Same accounting as
    # c2s (final state + metrics + events, no per-step state in memory); never the headline
    "c3s": dict(R=512, E=64, b_alg=16.0, stored=8.0, config=3, sliced=True),
    "c5": dict(R=1024, E=256, b_alg=154.0, stored=112.0, config=4),
    # the c5 crowd with ONE PID car per scenario among the 255 pedestrians: the crowd kernel with riders (the car's poses come
    # from the controller pre-pass); SG_CROWD_RIDERS=0 runs the general pedestrian variant instead -- the cliff, measured
    "c5mix": dict(R=1024, E=256, b_alg=154.0, stored=112.0, config=4, mix=True),
    # the c5 crowd as the reference actually runs crowds (pedestrian/sensor.py:50-51 refuses to run without a road network,
    # examples/crowds.py:149-205): on a pavement with four buildings, routes along the streets, the boundary forces of
    # social_force.py:86-104 as a phase of the crowd kernel
    "c5roads": dict(R=1024, E=256, b_alg=154.0, stored=112.0, config=4, roads=True),
    # the c3 batch with the RSSDistances state callback (metrics/rss/callback.py:58-128) after the reset and after every step,
    # inside the rollout kernel: + one record per entity-step (code 4 B, safe lateral / longitudinal distance 16 B)
    "c3rss": dict(R=4096, E=64, b_alg=134.0, stored=72.0, config=2, rss=True),
}


def effective_cpus():
    """os.cpu_count() capped by the cgroup CPU quota (the GPU boxes show 256 logical CPUs under a quota of 16)."""
    from scenario_gym_amd.packing import effective_cpus as f

    return f()


def cpu_baseline(workload, seconds_budget=20.0):
    """The CPU oracle (a port of the reference's algorithm, NOT the thing measured) on a bounded
    sample of the same workload: one scenario per host thread, the same E, shortened T."""
    import concurrent.futures as cf

    from oracle import oracle as O
    from scenario_gym_amd import synthetic
    from scenario_gym_amd.packing import unpack_scenario

    O.build()
    O.lib()
    cores = effective_cpus()
    E, T = workload["E"], workload["T"]
    rss = bool(workload.get("rss"))
    net = None
    if workload.get("crowd"):
        t_sample, n_scen = min(T, 1000), min(max(cores, 1) * 16, 1024)
        if workload.get("roads"):
            packed, net, _ = synthetic.make_crowd_roads(n_scen, E, n_steps=t_sample)
        else:
            packed = synthetic.make_crowd(n_scen, E, n_steps=t_sample)
    elif rss:  # (the oracle's callback runs over the recorded poses of its rollout)
        t_sample, n_scen = min(T, 500), min(max(cores, 1) * 16, 1024)
        packed = synthetic.make_batch(n_scen, E, n_steps=t_sample, ego_kind=workload["ego_kind"])
    else:
        t_sample, n_scen = T, min(max(cores, 1) * 64, 4096)
        packed = synthetic.make_batch(n_scen, E, n_steps=t_sample, ego_kind=workload["ego_kind"])
    scen = [unpack_scenario(packed, r) for r in range(n_scen)]

    def one(s):
        o = O.rollout(s["knot_off"], s["knots"], s["bbox"], s["etype"], s["kind"], s["ego"], s["t0"],
                      s["length"], workload["dt"], ctrl=s["ctrl"], max_steps=t_sample, record=rss,
                      route_off=s.get("route_off"), routes=s.get("routes"), road=net)
        if rss:
            O.rss_rollout(o, s["bbox"], s["ego"])
        return o["n_steps"]

    one(scen[0])  # warm
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(cores) as ex:  # ctypes releases the GIL: real parallelism
        steps = list(ex.map(one, scen))
    dt = time.perf_counter() - t0
    out = {
        "value": float(sum(steps)) * E / dt, "unit": "entity-steps/s", "cores": cores, "kind": "port",
        "sample": f"{n_scen} scenarios x {E} entities x {t_sample} steps of the same seeded family, "
                  f"{cores} threads (= the CPUs the cgroup quota grants of {os.cpu_count()} logical), C oracle "
                  f"(oracle/sgym_oracle.c), {dt:.1f}s",
    }
    # the real reference cannot travel to the GPU box: its numbers from the build container (tools/time_reference.py)
    refs = {}
    for key, name in (("reference", "reference_cpu_c3.json"), ("reference_default_agents", "reference_cpu.json")):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            r = json.load(f)
        refs[key] = {
            "value": r["entity_steps_per_s_box"], "per_core": r["entity_steps_per_s_per_core"], "cores": r["processes"],
            "unit": "entity-steps/s", "config": r.get("config", "other"), "ego": r.get("ego", "replay"),
            "entities": r["entities"], "sim_steps": r["sim_steps"], "scenarios_per_process": r["scenarios_per_process"],
            "collision_metric": bool(r.get("collision_metric", False)),
            "note": f"driskai/scenario_gym v0.3.1 itself, imported with stand-ins for the absent lxml / shapely; "
                    f"{r['processes']} single-threaded processes x {r['scenarios_per_process']} scenarios x {r['entities']} entities x "
                    f"{r['sim_steps']} steps, ego = {r.get('ego', 'replay')}, 3 ego metrics, no CollisionMetric (GEOS is absent; the "
                    f"stand-in's exact-rational SAT says nothing about it); timed in the 8-core build container "
                    f"(tools/time_reference.py, profiles/{name})",
        }
    if workload.get("crowd") or rss:  # (the reference was timed on the c3 / default-agent vehicle batches only)
        refs = {k + "_c3_batch": v for k, v in refs.items()}
    out.update(refs)
    return out


def kernel_name(E, crowd, controlled, rss=False, mix=False):
    """Entry point the library launches for this shape (sgym_hip.hip launch_variant): tile lanes G, wavefronts per tile;
    `controlled`: the batch has PID / vehicle agents (their pre-pass table is replayed by rollout_kernel_tab)."""
    G, WV = min(64, max(4, 1 << (E - 1).bit_length())), (1 if E <= 64 else 2 if E <= 128 else 4)
    if rss:  # (controlled lanes on the pre-pass table: the variant without in-kernel controllers)
        if WV == 1 and controlled and os.environ.get("SG_RSS_TAB", "1") != "0":
            return f"sg::rollout_kernel_rss_tab{'q' if os.environ.get('SG_QUEUE', '1') != '0' else ''}<{G}>"
        return f"sg::rollout_kernel_rss<{G}, {WV}>"
    if crowd:
        if G == 64:  # (a crowd with riders -- lanes of other kinds on a pre-pass table -- has its own entry point)
            return f"sg::rollout_kernel_crowd_riders<{WV}>" if mix and os.environ.get("SG_CROWD_RIDERS", "1") != "0" else \
                (f"sg::rollout_kernel<64, {WV}, true, false>" if mix else f"sg::rollout_kernel_crowd<{WV}>")
        return f"sg::rollout_kernel<{max(G, 16)}, {WV}, true, false>"
    if WV == 1 and controlled:  # (the synthetic batches are planar: z = pitch = roll = +0.0 in every knot)
        q = "q" if os.environ.get("SG_QUEUE", "1") != "0" else ""  # rollout_kernel_tabq*: the persistent launch (sgym_queue.hpp)
        return f"sg::rollout_kernel_tab{q}_planar<{G}>" if os.environ.get("SG_PLANAR", "1") != "0" else f"sg::rollout_kernel_tab{q}<{G}>"
    return f"sg::rollout_kernel<{G}, {WV}, false, true>"


def committed_profile(workload, kind, R, E, T=None):
    """profiles/latest_<workload>_<kind>.json (written by tools/profile_round.sh on the GPU box and committed), or None
    when it was taken on another shape or on other kernel sources than the ones this run was built from."""
    import scenario_gym_amd._lib as L

    path = os.path.join(ROOT, "profiles", f"latest_{workload}_{kind}.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        rec = json.load(f)
    if (rec.get("scenarios"), rec.get("entities")) != (R, E) or (T is not None and rec.get("sim_steps") != T):
        return None
    if rec.get("src_sha16") != L.source_sha16():
        return None
    return rec


def counted_flops(workload, E, T):
    """profiles/flops_<key>.json (tools/count_flops.py: the counter build of the CPU oracle over scenarios of this very
    batch, full horizon), or None when it was counted on another shape."""
    key = FLOPS_OF.get(workload)
    path = os.path.join(ROOT, "profiles", f"flops_{key}.json") if key else None
    if not path or not os.path.exists(path):
        return None
    with open(path) as f:
        rec = json.load(f)
    if (rec.get("entities"), rec.get("sim_steps")) != (E, T):
        return None
    rec["file"] = f"profiles/flops_{key}.json"
    rec["exact"] = key == workload or (workload, key) in (("c3s", "c3"), ("c2s", "c2"))
    return rec


def valu_peak(device):
    """tools/valu_peak.hip run live on this GPU: peak vector-ALU wavefront-instructions / s (and what one dependent fp64
    chain per wavefront reaches at the rollout kernels' two wavefronts per SIMD)."""
    path = os.path.join(ROOT, "scenario_gym_amd", "lib", "libvalu_peak.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    out = (ctypes.c_double * 4)()
    if lib.valu_peak_measure(ctypes.c_int(device), ctypes.c_int(2), out) != 0:
        return None
    return {"instr_per_s": out[0], "fp64_tflops": out[1], "dependent_chain_instr_per_s_2_waves": out[2], "cus": int(out[3])}


def timed_passes(one_pass, steps, warmup, dist, sync):
    """W untimed + K timed passes bracketed by barrier + device synchronisation; returns the local wall time, the
    entity-steps this rank processed and the per-pass kernel statistics."""
    for _ in range(warmup):
        one_pass()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    stats, ent_steps = [], 0
    for _ in range(steps):
        n, ms = one_pass()
        stats.append(ms)
        ent_steps += n
    sync()
    if dist is not None:
        dist.barrier()
    return time.perf_counter() - t0, ent_steps, stats


def collect(elapsed, ent_steps, dist):
    """max-over-ranks time, whole-job entity-steps, and every rank's own throughput (rank 0 gets the list)."""
    import numpy as np

    from scenario_gym_amd import distributed as D

    per_rank = D.gather_rows(np.array([[ent_steps / elapsed]]), dist)
    return D.max_over_ranks(elapsed, dist), D.sum_over_ranks(float(ent_steps), dist), \
        (None if per_rank is None else [float(v) for v in per_rank[:, 0]])


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` outside a launcher: start the N ranks ourselves (one process per GPU, torch.distributed.run,
    rendezvous on 127.0.0.1) as a CHILD of this process -- which has not imported torch nor touched HIP -- and relay rank
    0's JSON line and the exit code.  Never exec, never spawn after a GPU call."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in proc.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1])
        sys.stdout.flush()
    return proc.returncode if (proc.returncode != 0 or lines) else 1


def load_factory(spec):
    """--engine-factory module:function (tests only: a stand-in engine so that the launch / dispatch / collection path runs
    on a CPU box); the function is called as f(R, first_scenario, E, T) and returns an engine-like object."""
    import importlib

    mod, fn = spec.split(":")
    return getattr(importlib.import_module(mod), fn)


def _e2e_make_files(args):
    """(not timed) one worker's share of the synthetic OpenSCENARIO directory."""
    root, lo, hi, n_entities, n_vertices, duration = args
    import numpy as np

    from scenario_gym_amd import xosc_write as W

    paths = []
    for i in range(lo, hi):
        rng = np.random.default_rng([20240807, i])
        p = os.path.join(root, "Scenarios", f"s{i:05d}.xosc")
        W.write_scenario(p, W.synthetic_entities(rng, n_entities, n_vertices, duration=duration, extent=60.0))
        paths.append(p)
    return paths


def _e2e_load(path):
    from scenario_gym_amd.xosc import load_scenario_file

    return load_scenario_file(path, relabel=True)


def _e2e_load_pack(arg):
    """One worker task: import a handful of files and pack them (default agents) -- a few arrays travel back, not objects."""
    from scenario_gym_amd.packing import load_and_pack, packed_to_shm

    paths, E = arg
    return packed_to_shm(load_and_pack(paths, E, relabel=True))  # (the knots through shared memory, not through pickle)


def run_e2e(args):
    """--workload e2e: files -> metrics, the one place the reference published a number ("around 43 scenarios
    per second at around 400x realtime", scenario-gym.pdf IV-A, with hazard / RSS metrics; reference pipeline
    manager.py:240-283).  A generated directory of OpenSCENARIO files (Argoverse-like: 40 entities, 11 s at 10 Hz) ->
    import_scenario (native scan, a pool of processes) -> pack -> sg_upload -> rollout with the RSSDistances callback +
    CollisionMetric (classified) + 3 ego metrics + RSS -> get_metrics(); the import of chunk k + 1 runs under the device
    work of chunk k.  Not the headline."""
    import concurrent.futures as cf
    import multiprocessing
    import shutil
    import tempfile

    import numpy as np

    import scenario_gym_amd as sga
    import scenario_gym_amd._lib as L
    from scenario_gym_amd import metrics as M
    from scenario_gym_amd import xosc_write as W

    n_files, E, nv, duration, dt = args.files, 40, 111, 11.0, 1.0 / 30.0
    cores = effective_cpus()
    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (n_files * 600_000 + (1 << 30)) else None
    root = tempfile.mkdtemp(prefix="sgym_e2e_", dir=base)
    ctx = multiprocessing.get_context("spawn")
    try:
        W.write_catalog(os.path.join(root, "Catalogs"))
        os.makedirs(os.path.join(root, "Scenarios"))
        t0 = time.perf_counter()
        with cf.ProcessPoolExecutor(cores, mp_context=ctx) as ex:
            shares = [(root, n_files * w // cores, n_files * (w + 1) // cores, E, nv, duration) for w in range(cores)]
            paths = [p for ps in ex.map(_e2e_make_files, shares) for p in ps]
        gen_s = time.perf_counter() - t0
        n_bytes = sum(os.path.getsize(p) for p in paths[:16]) / 16 * len(paths)
        chunk = min(args.chunk, n_files)
        chunks = [paths[i:i + chunk] for i in range(0, n_files, chunk)]
        stages = dict(ingest=0.0, merge=0.0, upload=0.0, device=0.0, readback=0.0)
        all_metrics = []
        SUB = int(os.environ.get("SG_E2E_SUB", "32"))  # files per worker task

        def factory():
            return [M.EgoAvgSpeed(), M.EgoMaxSpeed(), M.EgoDistanceTravelled(), M.CollisionMetric(), M.RSS()]

        def tasks(ch):
            return [(ch[i:i + SUB], E) for i in range(0, len(ch), SUB)]

        def merged(fut):
            """(a helper thread) wait for the import of a chunk, then merge the workers' arrays: one copy out of their
            shared-memory segments; numpy's copies release the GIL, so this runs under the main thread's upload of the chunk before"""
            from scenario_gym_amd.packing import merge_packed_shm, release_shm

            t = time.perf_counter()
            parts = []
            try:
                for q in fut:
                    parts.append(q)
            except BaseException:
                release_shm(parts)  # (a worker failed: the segments received so far would stay in /dev/shm)
                raise
            t1 = time.perf_counter()
            packed = merge_packed_shm(parts)  # (removes its segments itself, also when it fails)
            return packed, t1 - t, time.perf_counter() - t1

        def device_part(gym, packed):
            t = time.perf_counter()
            gym.set_packed(packed)                         # sg_upload (+ the reset launch); the engine is reused
            stages["upload"] += time.perf_counter() - t
            t = time.perf_counter()
            gym.rollout()
            gym.engine.synchronize()
            stages["device"] += time.perf_counter() - t
            t = time.perf_counter()
            out = gym.get_metrics()
            stages["readback"] += time.perf_counter() - t
            return out

        with cf.ProcessPoolExecutor(cores, mp_context=ctx) as ex:
            from scenario_gym_amd.packing import packed_from_shm as _drop

            for q in ex.map(_e2e_load_pack, [([p], E) for p in paths[:cores]]):   # (workers started and warm: imports, catalog cache)
                _drop(q)
            gym = sga.BatchedScenarioGym(timestep=dt, state_callbacks=[M.RSSDistances()], metrics=factory, event_capacity=16)
            # warm, not timed: library load, the engine of the chunk shape with its allocations (the RSS line-test queue alone
            # is GiBs), first launches
            device_part(gym, merged(ex.map(_e2e_load_pack, tasks(chunks[0])))[0])
            for k in stages:
                stages[k] = 0.0
            stages["main_waits_for_chunk"] = 0.0
            import gc

            import torch

            # the interpreter's cyclic collector: a full pass walks every list of entity names and metric objects made so far
            # (42 ms every tenth chunk, measured) and finds nothing -- the sweep builds no cycles.  What exists is frozen and
            # the young generation made large; reference counting still frees every chunk's objects as they go.
            gc.collect()
            gc.freeze()
            gc_thresholds = gc.get_threshold()
            gc.set_threshold(200_000, 50, 1000)
            torch.cuda.synchronize()
            t_all = time.perf_counter()
            with cf.ThreadPoolExecutor(1) as helper:
                # the import pool works two chunks ahead, the helper thread merges one chunk ahead, the main thread uploads
                # and runs the device
                futs = [ex.map(_e2e_load_pack, tasks(chunks[c])) for c in range(min(2, len(chunks)))]
                nxt = helper.submit(merged, futs[0])
                for c in range(len(chunks)):
                    t = time.perf_counter()
                    packed, t_ing, t_mer = nxt.result()
                    stages["main_waits_for_chunk"] += time.perf_counter() - t
                    stages["ingest"] += t_ing
                    stages["merge"] += t_mer
                    if c + 2 < len(chunks):
                        futs.append(ex.map(_e2e_load_pack, tasks(chunks[c + 2])))
                    if c + 1 < len(chunks):
                        nxt = helper.submit(merged, futs[c + 1])
                    all_metrics.extend(device_part(gym, packed))
            wall = time.perf_counter() - t_all
            gc.set_threshold(*gc_thresholds)
            gc.unfreeze()
            gym.close()
        assert len(all_metrics) == n_files
        steps = int(np.ceil(duration / dt))
        line = {
            "metric": "scenarios/s (OpenSCENARIO files -> metrics, end to end)", "value": n_files / wall, "unit": "scenarios/s",
            "n_gpus": 1, "steps": 1, "warmup": 1, "ms_per_step": wall * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "x_realtime": n_files * duration / wall,
            "entity_steps_per_s": n_files * E * steps / wall,
            "stage_seconds": {k: round(v, 4) for k, v in stages.items()},
            "stage_note": "three stages side by side: the import pool (native scan + normalisation + packing as arrays, two chunks "
                          "ahead; its knots come back through shared memory), a helper thread (ingest = its wait for the pool, "
                          "merge = one copy of the workers' rows into the chunk's batch) and the main thread (upload = sg_upload + "
                          "reset, device = sg_rollout incl. the RSS callback, readback = get_metrics; main_waits_for_chunk = its "
                          "idle time)",
            "config": {"name": "e2e", "workload": f"{n_files} generated OpenSCENARIO files ({E} entities, {nv} vertices = {duration:g} s at 10 Hz, "
                                                  f"~{n_bytes / n_files / 1e3:.0f} kB each) -> import_scenario (native scan, {cores} processes) -> "
                                                  f"pack -> sg_upload -> rollout at dt = 1/30 ({steps} steps) with RSSDistances + CollisionMetric "
                                                  f"(classified) + EgoAvgSpeed / MaxSpeed / DistanceTravelled + RSS -> get_metrics, chunks of {chunk}",
                       "files": n_files, "entities": E, "sim_steps": steps, "timestep": dt, "chunk": chunk, "host_processes": cores,
                       "bytes": int(n_bytes), "generate_seconds_not_timed": round(gen_s, 2)},
            "published_reference": {"value": 43.0, "unit": "scenarios/s", "x_realtime": 400.0,
                                    "note": "BASELINE.md 1 / scenario-gym.pdf IV-A: the reference's own figure on its hardware and the "
                                            "Argoverse recordings, with hazard / RSS metrics -- other data, other machine: context, not a ratio"},
            "sample_metrics": all_metrics[0],
        }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = e2e_cpu_baseline(paths, dt, steps, cores)
        print(json.dumps(line, default=float))
        return line
    finally:
        shutil.rmtree(root, ignore_errors=True)


def e2e_cpu_baseline(paths, dt, steps, cores):
    """The same pipeline with the CPU oracle in place of the device (a port, not the reference): import -> oracle rollout with
    its per-step RSS callback -> metrics, one scenario per thread, on a bounded sample."""
    import concurrent.futures as cf

    from oracle import oracle as O
    from scenario_gym_amd.packing import pack_scenarios, unpack_scenario
    from scenario_gym_amd.xosc import load_scenario_file

    O.build()
    O.lib()
    sample = paths[: max(cores * 4, 16)]

    def one(p):
        sc = load_scenario_file(p, relabel=True)
        packed, _ = pack_scenarios([sc])
        s = unpack_scenario(packed, 0)
        o = O.rollout(s["knot_off"], s["knots"], s["bbox"], s["etype"], s["kind"], s["ego"], s["t0"], s["length"], dt,
                      ctrl=s["ctrl"], max_steps=steps + 8, record=True)
        O.rss_rollout(o, s["bbox"], s["ego"])
        return o["n_steps"]

    one(sample[0])
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(cores) as ex:
        n = list(ex.map(one, sample))
    dtm = time.perf_counter() - t0
    return {"value": len(sample) / dtm, "unit": "scenarios/s", "cores": cores, "kind": "port",
            "sample": f"{len(sample)} of the files, {cores} threads: import_scenario + oracle/sgym_oracle.c rollout ({int(sum(n) / len(n))} steps) with "
                      f"its RSS callback driven step by step from Python + metrics, {dtm:.1f}s"}


def main(argv=None, make_engine=None, emit=True):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scenarios", type=int, default=None, help="scenarios per GPU (default: the workload's own, 4096 for c3)")
    ap.add_argument("--entities", type=int, default=None)
    ap.add_argument("--sim-steps", type=int, default=10000)
    ap.add_argument("--ego", default="pid", choices=["pid", "replay"])
    ap.add_argument("--workload", default="c3", choices=["c3", "c2", "c2s", "c5", "c5mix", "c5roads", "c3rss", "c3s", "e2e"],
                    help="BASELINE.json configs: c3 = 4096x64 PID ego (default, the headline), c2 = 256x16 replay (state of "
                         "every step materialised), c5 = 1024x256 social-force crowd; c2s = the c2 batch through the "
                         "time-sliced replay path (final state + metrics + events only: a separate mode, never the headline); c3rss = the c3 "
                         "batch with the RSSDistances callback after every step")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="what `value` is: strong (the default with more than one rank) = --scenarios in TOTAL, split evenly over "
                         "the ranks -- BASELINE.json's metric and configs[3] as written: 4096 scenarios x 64 entities at 1 / 2 / 4 / 8 "
                         "GPUs = 512 per GPU at 8; weak = --scenarios per GPU.  With one rank the two are the same batch (reported "
                         "as \"weak\": per-GPU work is what it is).  With N > 1 the other one is timed too and reported under its own key")
    ap.add_argument("--no-configs", action="store_true",
                    help="c3 on one GPU: do not time the other single-GPU BASELINE configs (c2, c5) after the headline's timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", type=int, default=None,
                    help="after the timed passes (outside the timed region) re-run this many scenarios spread over the batch "
                         "through the CPU oracle for the full horizon and compare the final state / metric rows / events bit "
                         "for bit; a mismatch makes the exit code non-zero.  Default 16 (c5, c3rss: 4); 0 = off")
    ap.add_argument("--engine-factory", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--require-queue", action="store_true",
                    help="exit non-zero (code 4) unless every rank ran the table path as the one persistent launch "
                         "(roofline.schedule; without the flag a rank that fell back to chunk launches only marks the line \"degraded\")")
    ap.add_argument("--files", type=int, default=4096, help="e2e: OpenSCENARIO files to generate and run")
    ap.add_argument("--chunk", type=int, default=512, help="e2e: scenarios per device batch")
    ap.add_argument("--ped-noise", default="off", choices=["off", "device"],
                    help="c5: SocialForce noise terms (social_force.py:106-114): off = std 0 (parity runs), device = the "
                         "reference's default std with the counter-based device RNG")
    args = ap.parse_args(argv)
    if args.workload == "e2e" and (args.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1):
        raise SystemExit("--workload e2e is a one-GPU pipeline (its host side is the bottleneck); run one per GPU instead")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and make_engine is None:
        # not under a launcher: be the launcher (before torch / HIP are touched in this process)
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:] if argv is None else argv))
    if args.engine_factory and make_engine is None:
        factory = load_factory(args.engine_factory)

        def make_engine(R, first):
            return factory(R, first, args.entities, args.sim_steps)

    if args.workload == "e2e":
        return run_e2e(args)

    import numpy as np

    import scenario_gym_amd._lib as L
    from scenario_gym_amd import distributed as D
    from scenario_gym_amd import synthetic

    wl = WORKLOADS[args.workload]
    args.scenarios = args.scenarios or wl["R"]  # (explicit --scenarios / --entities: size sweeps of the same family)
    args.entities = args.entities or wl["E"]
    if args.workload in ("c2", "c2s"):
        args.ego = "replay"
    crowd = args.workload in ("c5", "c5mix", "c5roads")
    rank, world, local_rank, dist = D.init()
    if args.scaling is None:
        args.scaling = "strong" if world > 1 else "weak"
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    dt = 1.0 / 30.0
    ego_kind = L.KIND_AGENT_PID if args.ego == "pid" else L.KIND_AGENT_REPLAY
    # dispatch: rank 0 broadcasts the run configuration (RCCL); each rank generates exactly its own shard
    R_arg, E, T, ego_kind, seed = D.dispatch_config(
        [args.scenarios, args.entities, args.sim_steps, ego_kind, synthetic.SEED], dist)
    if R_arg % world:
        raise SystemExit(f"{R_arg} scenarios do not split evenly over {world} ranks")
    shapes = {"weak": R_arg, "strong": R_arg // world}  # scenarios per rank

    live = make_engine is None
    if live:
        import torch

        import scenario_gym_amd as sga

        def sync():
            torch.cuda.synchronize()

        def make_engine(R, first, sliced=None):
            road = None
            if crowd and wl.get("roads"):
                packed, net, net_of = synthetic.make_crowd_roads(R, E, n_steps=T, timestep=dt, seed=seed, first_scenario=first)
                road = (net, net_of)
            elif crowd and wl.get("mix"):
                packed = synthetic.make_crowd_with_car(R, E, n_steps=T, timestep=dt, seed=seed, first_scenario=first)
            elif crowd:
                packed = synthetic.make_crowd(R, E, n_steps=T, timestep=dt, seed=seed, first_scenario=first)
            else:
                packed = synthetic.make_batch(R, E, n_steps=T, timestep=dt, ego_kind=ego_kind, seed=seed, first_scenario=first)
            kw = {}
            if crowd and args.ped_noise == "device":
                kw["social_force"] = dict(std_lon=0.1, std_lat=0.1, noise="device")
            eng = sga.RolloutEngine(R, E, timestep=dt, terminal_conditions=["max_length"], event_capacity=64,
                                    device=local_rank, **kw)
            # (every workload but c2s / c3s materialises the state of every step)
            eng.set_slicing(bool(wl.get("sliced")) if sliced is None else sliced)
            if wl.get("rss"):
                eng.set_rss(True)
            eng.upload(packed)
            if road:
                eng.set_road_networks([road[0]], road[1])
            eng._bench_packed = packed  # (kept for the oracle check after the timed passes)
            eng._bench_road = road
            return eng
    else:  # tests drive the dispatch / timing / collection code with a stand-in engine on CPU
        def sync():
            pass

    n_verify = args.verify if args.verify is not None else (4 if (crowd or wl.get("rss")) else 16)

    def verify(eng, K):
        """NOT timed: the state the last timed pass left against the CPU oracle, full horizon, K scenarios of this rank."""
        from oracle import check

        noise_of = None
        if crowd and args.ped_noise == "device":
            noise_of = lambda r: dict(mode="device", std_lon=0.1, std_lat=0.1, seed=0, scenario_index=r)  # noqa: E731
        t0 = time.perf_counter()
        v = check.verify_engine(eng, eng._bench_packed, dt, T, K=K, event_cap=64, ped=crowd, rss=bool(wl.get("rss")),
                                noise_of=noise_of, threads=effective_cpus(),
                                road_of=(lambda r: eng._bench_road[0]) if getattr(eng, "_bench_road", None) else None)
        v["seconds"] = round(time.perf_counter() - t0, 2)
        v["what"] = ("after the timed passes, outside the timed region: the device state left by the last timed rollout vs "
                     "oracle/sgym_oracle.c run over the full horizon, bit for bit")
        return v

    def measure(R, sliced=None):
        pipes, kernels = [], []
        first = rank * R  # rank r owns scenarios [r R, (r + 1) R) of the seeded family (chunk-aligned: R % 64 == 0 or 1 rank)
        eng = make_engine(R, first) if sliced is None else make_engine(R, first, sliced)

        def one_pass():
            eng.rollout_async(T, do_reset=True)
            eng.synchronize()
            rows, _ = eng.metrics()
            # collection: every rank's per-replica metric rows to rank 0 (RCCL gather)
            D.gather_rows(np.stack([rows["ego_avg_speed"], rows["ego_max_speed"], rows["ego_distance_travelled"],
                                    rows["n_collisions"].astype(np.float64), rows["n_steps"].astype(np.float64)],
                                   axis=1), dist)
            n_launch, launch_ms = eng.last_launch_stats()
            gross = eng.last_launch_gross_ms() if hasattr(eng, "last_launch_gross_ms") else launch_ms
            pipes.append(eng.schedule_info() if hasattr(eng, "schedule_info") else None)
            kernels.append(eng.last_kernel() if hasattr(eng, "last_kernel") else None)
            return int(rows["n_steps"].sum()) * E, (eng.last_kernel_ms(), n_launch, launch_ms, gross)

        elapsed, ent_steps, stats = timed_passes(one_pass, args.steps, args.warmup, dist, sync)
        ver = None
        if n_verify > 0 and getattr(eng, "_bench_packed", None) is not None:
            ver = verify(eng, max(2, -(-n_verify // world)))
        eng.close()
        worst, total, per_rank = collect(elapsed, ent_steps, dist)
        if ver is not None and dist is not None:  # every rank checks scenarios of its own shard
            ver["equal"] = D.sum_over_ranks(0.0 if ver["equal"] else 1.0, dist) == 0.0
            ver["scenarios"] = int(D.sum_over_ranks(float(ver["scenarios"]), dist))
        # every rank's launch schedule to rank 0: [schedule of the timed passes (min over them), chunks, ring, grid, pre-pass
        # wavefronts, blocks, SIMDs, launches per call]
        pi = pipes[-1] if pipes and pipes[-1] else None
        sched = D.gather_rows(np.array([[min(q["schedule"] for q in pipes), pi["chunks"], pi["ring"], pi["grid"], pi["ctl_waves"],
                                         pi["blocks"], pi["simds"], pi["launches"]]] if pi else [[0.0] * 8], np.float64), dist)
        return dict(elapsed=worst, total=total, per_rank=per_rank, ent_steps=ent_steps, stats=stats, R=R, verified=ver,
                    sched=None if sched is None else sched.tolist(), kernel=kernels[-1] if kernels else None)

    main_run = measure(shapes[args.scaling])
    other_run = None
    if world > 1:
        other = "strong" if args.scaling == "weak" else "weak"
        other_run = (other, measure(shapes[other]))

    sliced_run = None
    if world > 1 and live and args.workload == "c3":
        # BASELINE.json configs[3] read literally leaves each GPU a shard that cannot fill it (512 wavefronts on 1024 SIMDs);
        # the time-sliced path (workload c3s: its own accounting, final state + metrics + events) is what such a shard should
        # run through: timed beside the step-materialised figure, never instead of it
        sliced_run = measure(shapes["strong"], sliced=True)

    line = None
    if rank == 0:
        m = main_run
        R = m["R"]
        # dominant kernel = the rollout kernel.  A long rollout is cut into chunks of steps (one launch each, so that the
        # controller pre-pass of the next chunk overlaps it): per-launch units and duration are averages over the
        # launches of the timed passes, each launch timed with its own HIP event pair on its stream.  Large batches with
        # controlled agents run as two halves on two streams (launch_rollout): two launches of the same kernel are then in
        # flight, and the time a launch is charged is its share of the time at least one of them was running -- the union
        # of the launches' intervals / launches (`kernel_ms`); `kernel_ms_gross` is the plain average duration of a launch,
        # the figure a kernel trace shows, `launch_overlap` their ratio (1.0: one launch at a time).
        n_launches = sum(k[1] for k in m["stats"])
        per_launch = m["ent_steps"] / n_launches
        avg_ms = sum(k[2] for k in m["stats"]) / n_launches
        gross_ms = sum(k[3] for k in m["stats"]) / n_launches
        rollout_ms = sum(k[0] for k in m["stats"]) / len(m["stats"])  # everything one sg_rollout enqueues
        b_alg = wl["b_alg"]
        achieved = per_launch * b_alg / (avg_ms * 1e-3) / 1e9
        launches_per_rollout = n_launches / args.steps
        # measured HBM traffic and instruction mix: only from a committed profile of THESE kernel sources on THIS shape
        hbm = committed_profile(args.workload, "hbm_traffic", R, E, T)
        traffic = hbm["hbm_bytes_per_rollout"] / launches_per_rollout if hbm else None
        sq = committed_profile(args.workload, "pmc_sq", R, E)
        peak = valu_peak(local_rank) if live else None
        secondary = None
        if sq and peak:
            EP = E if E > 64 else max(4, 1 << (E - 1).bit_length())
            wave_steps_per_s = (m["ent_steps"] / E * EP / 64) / (sum(k[2] for k in m["stats"]) * 1e-3)
            pws = sq["per_wave_step"]
            valu = pws["SQ_INSTS_VALU"] * wave_steps_per_s
            secondary = {
                "bound": "valu_issue", "unit": "wavefront-instructions/s", "peak": peak["instr_per_s"], "achieved": valu,
                "frac": valu / peak["instr_per_s"], "valu_per_wave_step": pws["SQ_INSTS_VALU"],
                "salu_per_wave_step": pws["SQ_INSTS_SALU"],
                "all_instr_frac": (pws["SQ_INSTS_VALU"] + pws["SQ_INSTS_SALU"] + pws.get("SQ_INSTS_LDS", 0.0)) *
                                  wave_steps_per_s / peak["instr_per_s"],
                "peak_fp64_tflops": peak["fp64_tflops"],
                "dependent_chain_peak": peak["dependent_chain_instr_per_s_2_waves"],
                "profile": f"profiles/latest_{args.workload}_pmc_sq.json ({sq.get('sim_steps')} steps)",
                "note": "peak: tools/valu_peak.hip run in this process; achieved: SQ_INSTS_VALU per wavefront-step of the "
                        "committed rocprofv3 --pmc pass x wavefront-steps / s of this run's rollout-kernel launches",
            }
        elif peak:
            secondary = {"bound": "valu_issue", "unit": "wavefront-instructions/s", "peak": peak["instr_per_s"],
                         "achieved": None, "frac": None, "peak_fp64_tflops": peak["fp64_tflops"],
                         "note": "no committed SQ profile of these kernel sources on this shape (profiles/latest_*_pmc_sq.json)"}
        # the dominant kernel: what the library says it launched (sg_last_kernel); a stand-in engine has nothing to say and gets
        # the name the shape implies
        kname = m.get("kernel") or (
            f"sg::rollout_kernel_slice{'_tab' if ego_kind == L.KIND_AGENT_PID else ''}<{min(64, max(4, 1 << (E - 1).bit_length()))}>" if wl.get("sliced") else
            kernel_name(E, crowd, ego_kind == L.KIND_AGENT_PID, bool(wl.get("rss")), bool(wl.get("mix"))))
        # The contract's HBM model (the step-materialised state as compulsory writes) does not bind this design: the
        # rows are rewritten in place every step and live in L2, so it can pass 1.  Kept as a secondary figure.
        hbm_contract = {
            "bound": "hbm", "unit": "GB/s", "achieved": achieved, "peak": HBM_PEAK_GBS, "frac": achieved / HBM_PEAK_GBS,
            "bytes_per_entity_step": b_alg, "stored_bytes_per_entity_step": wl["stored"], "traffic": traffic,
            "traffic_ratio": (traffic / (per_launch * b_alg)) if traffic else None,
            "note": "SURVEY 8d's algorithmic bytes x entity-steps per launch / kernel_ms over the 8 TB/s HBM peak.  NOT the binding "
                    "roof: the state rows are overwritten in place every step and stay in L2 / MALL (traffic = measured HBM bytes "
                    "per launch, calibrated counters), so this can exceed 1",
        }
        fl = counted_flops(args.workload, E, T)
        if fl:
            F = fl["flops_per_entity_step_total"]
            F0 = fl["flops_per_entity_step_without_pair_search"]
            tf = per_launch * F / (avg_ms * 1e-3) / 1e12
            tf0 = per_launch * F0 / (avg_ms * 1e-3) / 1e12
            roofline = {
                # primary roof: the fp64 vector ALU, numerator = ALGORITHMIC flops counted by the oracle's counter build on
                # scenarios of this very batch (not executed instructions: `valu_issue` below is the utilisation figure).
                # `achieved` / `frac` leave the pair search out: SURVEY 8d charges 6 flops per unordered pair of
                # present entities for it, the stripe-mask broad phase executes none of them (and where a broad phase runs
                # it is fp32) -- the figure with that charge is kept beside it, labelled
                "bound": "valu_fp64", "unit": "TFLOP/s", "achieved": tf0, "peak": FP64_VALU_PEAK_TFLOPS,
                "frac": tf0 / FP64_VALU_PEAK_TFLOPS,
                "flops_per_entity_step": F0, "flops_by_category": fl["flops_per_entity_step"], "flops_file": fl["file"],
                "flops_exact_for_this_workload": fl["exact"],
                "flops_per_entity_step_with_pair_search": F,
                "achieved_with_pair_search": tf, "frac_with_pair_search": tf / FP64_VALU_PEAK_TFLOPS,
                "peak_measured": peak["fp64_tflops"] if peak else None,
                "frac_of_measured_peak": (tf0 / peak["fp64_tflops"]) if peak and peak.get("fp64_tflops") else None,
                "traffic": traffic,
                "binding": "ordered_sum_latency" if wl.get("sliced") else "valu_issue",
                "note": "achieved = counted algorithmic fp64 flops per entity-step (profiles/flops_*.json: add/sub/mul/div/sqrt/"
                        "compare = 1, fma = 2, minimal formulation) WITHOUT the pair search x entity-steps per launch / kernel_ms; "
                        "*_with_pair_search adds SURVEY 8d's 6 flops per unordered pair of present entities, which the "
                        "stripe-mask broad phase does NOT execute; peak = 256 CUs x 4 SIMDs x 16 lanes x 2 x 2.4 GHz; "
                        "peak_measured = tools/valu_peak.hip run in this process; traffic = HBM bytes per launch (calibrated PMC counters)"
                        + ("" if fl["exact"] else "; the flops are those of the nearest counted workload (the RSS callback / the car "
                                                  "among the pedestrians are not in the count): a lower bound"),
            }
        else:  # no count for this shape: the contract's HBM figure stands in
            roofline = dict(hbm_contract)
            roofline["binding"] = "ordered_sum_latency" if wl.get("sliced") else "valu_issue"
        sched = m.get("sched")
        schedule = None
        if sched and any(r_[5] for r_ in sched):
            names = {0: "none", 1: "chunk_launches", 2: "persistent_queue"}
            schedule = {"per_rank": [names.get(int(r_[0]), "?") for r_ in sched], "chunks": int(sched[0][1]), "table_ring": int(sched[0][2]),
                        "wavefronts": int(sched[0][3]), "prepass_wavefronts": int(sched[0][4]), "blocks_per_rank": int(sched[0][5]),
                        "simds": int(sched[0][6]), "launches_per_rollout": int(sched[0][7]),
                        "note": "persistent_queue = the controller pre-pass and every chunk of the rollout in ONE launch, work items "
                                "(chunk, block) from a device-side counter (csrc/sgym_queue.hpp); no timing probe, no hardware-queue "
                                "dependence"}
            # the table kernels of a batch with controlled lanes are expected to run as the persistent launch
            if (kname.startswith("sg::rollout_kernel_tab") or kname.startswith("sg::rollout_kernel_rss_tab")) and any(int(r_[0]) != 2 for r_ in sched):
                schedule["degraded"] = True
                print(f"bench: DEGRADED launch schedule: {schedule['per_rank']} (SG_QUEUE=0, or the table ring could not be "
                      f"allocated); results are the same, throughput is not", file=sys.stderr)
        roofline.update({
            "kernel": kname, "kernel_ms": avg_ms, "kernel_ms_gross": gross_ms,
            "launch_overlap": gross_ms / avg_ms if avg_ms else None,
            "kernel_ms_note": "kernel_ms = union of the launches' HIP-event intervals / launches; kernel_ms_gross = plain average launch "
                              "duration, the figure rocprofv3 --kernel-trace --stats shows (equal: one launch at a time)",
            "launches_per_rollout": launches_per_rollout, "rollout_device_ms": rollout_ms,
            "entity_steps_per_launch": per_launch, "schedule": schedule,
            "valu_issue": secondary, "hbm_contract": hbm_contract if fl else None,
            "src_sha16": L.source_sha16(),
        })
        line = {
            "metric": "entity-steps/sec (batched rollout)",
            "value": m["total"] / m["elapsed"],
            "unit": "entity-steps/s",
            "n_gpus": world,
            "ranks": world if dist is None else dist.get_world_size(),
            "backend": "none" if dist is None else dist.get_backend(),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": m["elapsed"] / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "per_rank_value": m["per_rank"],
            "config": {
                "name": args.workload,
                "workload": (f"{R} scenarios x ({E - 1} pedestrians + 1 PID car) x {T} steps per GPU, PedestrianAgent + SocialForce / "
                             f"PIDAgent, all-pairs OBB collisions, CollisionMetric, terminal max_length (BASELINE.json configs[4] "
                             f"with a vehicle in the crowd)") if (crowd and wl.get("mix")) else
                            (f"{R} scenarios x {E} pedestrians x {T} steps per GPU on a road network (a pavement, four buildings; routes along "
                             f"the streets), PedestrianAgent + SocialForce with its boundary forces (radius 3 m, noise {args.ped_noise}) + "
                             "PedestrianController, all-pairs OBB collisions, CollisionMetric, terminal max_length (BASELINE.json "
                             "configs[4] as examples/crowds.py runs crowds)") if (crowd and wl.get("roads")) else
                            (f"{R} scenarios x {E} pedestrians x {T} steps per GPU, PedestrianAgent + SocialForce "
                             f"(radius 3 m, noise {args.ped_noise}) + PedestrianController, all-pairs OBB collisions, "
                             "CollisionMetric, terminal max_length (BASELINE.json configs[4])") if crowd else
                            (f"{R} scenarios x {E} entities x {T} steps per GPU, "
                             f"{'PIDAgent' if ego_kind == L.KIND_AGENT_PID else 'ReplayTrajectoryAgent'} ego + batch replay "
                             "others, all-pairs OBB collisions, CollisionMetric + EgoAvgSpeed/MaxSpeed/DistanceTravelled, "
                             f"terminal max_length (BASELINE.json configs[{wl['config']}])"
                             + ("; + RSSDistances state callback (metrics/rss/callback.py:58-128) after the reset and after "
                                "every step, inside the rollout kernel (rss_lines_kernel finishes the line tests after each "
                                "launch, inside the timed launches)" if wl.get("rss") else "")
                             + ("; TIME-SLICED replay path: final state + metrics + events, bit-identical to the step-by-step "
                                "kernel, the states of the intermediate steps are not written to memory" if wl.get("sliced") else "")),
                "scenarios_per_gpu": R, "entities": E, "sim_steps": T, "timestep": dt,
                "sharding": f"replicas x{world}, no data-path collective",
            },
            "roofline": roofline,
        }
        line["verified"] = m["verified"]
        if other_run:
            name, o = other_run
            line[name] = {"value": o["total"] / o["elapsed"], "ms_per_step": o["elapsed"] / args.steps * 1e3,
                          "scenarios_per_gpu": o["R"], "per_rank_value": o["per_rank"], "verified": o["verified"]}
        elif world == 1:
            # one rank: the weak and the strong shape are the same batch -- the block repeats `value` (no second run), so that
            # the N = 1 point of a scaling run carries the same keys as its N > 1 points
            other = "strong" if args.scaling == "weak" else "weak"
            line[other] = {"value": line["value"], "ms_per_step": line["ms_per_step"], "scenarios_per_gpu": R,
                           "per_rank_value": m["per_rank"], "verified": m["verified"],
                           "note": "one rank: the same batch as `value`, not measured twice"}
            if args.workload == "c3" and live:
                line["strong_sliced"] = {"value": None, "scenarios_per_gpu": R,
                                         "note": "one rank: the shard is the whole batch, which fills the chip -- the time-sliced "
                                                 "path engages only on shards of <= 1024 blocks (N >= 4 at 4096 scenarios)"}
        if sliced_run:
            o = sliced_run
            line["strong_sliced"] = {"value": o["total"] / o["elapsed"], "ms_per_step": o["elapsed"] / args.steps * 1e3,
                                     "scenarios_per_gpu": o["R"], "per_rank_value": o["per_rank"], "verified": o["verified"],
                                     "accounting": "workload c3s: time-sliced rollout, final state + metrics + events bit-identical "
                                                   "to the step-by-step path, the intermediate states are not written to memory"}
        if schedule and schedule.get("degraded"):
            line["degraded"] = "a rank ran the table path as chunk launches instead of the persistent launch (roofline.schedule)"
        # which engine produced the line (a stand-in injected by the CPU tests must never pass for the HIP library)
        line["engine"] = "scenario_gym_amd.RolloutEngine (libsgym_hip.so)" if live else (args.engine_factory or "injected stand-in (tests)")
        if live and not args.no_cpu_baseline:  # (rank 0 only, after the timed region; the other ranks wait at the teardown)
            line["cpu_baseline"] = cpu_baseline(dict(E=E, T=T, dt=dt, ego_kind=ego_kind, crowd=crowd, rss=bool(wl.get("rss")), roads=bool(wl.get("roads"))))
        if (live and world == 1 and args.workload == "c3" and not args.no_configs and emit
                and (R, E, T) == (WORKLOADS["c3"]["R"], WORKLOADS["c3"]["E"], 10000) and args.ego == "pid"):
            # the driver times ONE line: the other single-GPU BASELINE configs ride in it, each measured
            # the same way after the headline's timed region -- 3 timed passes, oracle-verified, its own roofline
            line["configs"] = {}
            for name in ("c2", "c5"):
                sub = main(["--workload", name, "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], emit=False)
                rf = sub["roofline"]
                line["configs"][name] = {
                    "value": sub["value"], "unit": sub["unit"], "ms_per_step": sub["ms_per_step"], "steps": sub["steps"],
                    "warmup": sub["warmup"], "workload": sub["config"]["workload"], "scenarios": sub["config"]["scenarios_per_gpu"],
                    "entities": sub["config"]["entities"], "sim_steps": sub["config"]["sim_steps"],
                    "roofline": {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "kernel_ms", "binding",
                                                        "flops_per_entity_step", "frac_with_pair_search")},
                    "verified": sub["verified"] and {k: sub["verified"][k] for k in ("equal", "scenarios", "steps")},
                }
        if emit:
            print(json.dumps(line))
    if dist is not None and (live or args.engine_factory):
        dist.destroy_process_group()
    failed = [(n, r) for n, r in (("value", main_run), ("strong / weak", other_run[1] if other_run else None), ("strong_sliced", sliced_run))
              if r and r["verified"] and not r["verified"]["equal"]]
    if failed:
        print(f"bench: the device state DIFFERS from the oracle in the `{failed[0][0]}` run of rank {rank}: {failed[0][1]['verified']['mismatches']}",
              file=sys.stderr)
        raise SystemExit(3)
    if args.require_queue and line is not None and (line.get("degraded") or not (line["roofline"].get("schedule") or {}).get("per_rank")):
        print("bench: --require-queue: not every rank ran the persistent table launch", file=sys.stderr)
        raise SystemExit(4)
    return line


if __name__ == "__main__":
    main()
