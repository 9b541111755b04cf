"""N>1 path on CPU: two gloo ranks shard a seeded batch, roll their shards out (with the oracle
standing in for the GPU, which is all a CPU-only box has) and collect the metric rows on rank 0."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from conftest import ROOT

R_TOTAL, E, STEPS = 128, 6, 40


def _rows_for(packed, oracle):
    from scenario_gym_amd.packing import unpack_scenario

    rows = []
    for r in range(packed.n_scenarios):
        s = unpack_scenario(packed, r)
        o = oracle.rollout(s["knot_off"], s["knots"], s["bbox"], s["etype"], s["kind"], s["ego"], s["t0"],
                           s["length"], 1 / 30, ctrl=s["ctrl"], max_steps=STEPS, record=False)
        rows.append([o["metric_ego_avg_speed"], o["metric_ego_max_speed"], o["metric_ego_distance_travelled"],
                     o["n_events"], o["n_steps"]])
    return np.array(rows, np.float64)


def _worker(rank, world, port, out_path):
    import sys

    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from oracle import oracle
    from scenario_gym_amd import distributed as D
    from scenario_gym_amd import synthetic

    rank, world, _, dist = D.init("gloo")
    cfg = D.dispatch_config([R_TOTAL, E, STEPS, 3] if rank == 0 else [0, 0, 0, 0], dist)
    assert cfg == [R_TOTAL, E, STEPS, 3]
    lo, hi = D.shard_bounds(cfg[0], rank, world)
    packed = synthetic.make_batch(hi - lo, cfg[1], n_steps=cfg[2], ego_kind=cfg[3], first_scenario=lo, extent=20.0)
    rows = D.gather_rows(_rows_for(packed, oracle), dist)
    total = D.sum_over_ranks(float(hi - lo), dist)
    worst = D.max_over_ranks(float(rank), dist)
    if rank == 0:
        np.savez(out_path, rows=rows, total=total, worst=worst)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_collection(tmp_path, oracle):
    from scenario_gym_amd import distributed as D
    from scenario_gym_amd import synthetic

    assert D.shard_bounds(4096, 0, 8) == (0, 512) and D.shard_bounds(4096, 7, 8) == (3584, 4096)
    assert [D.shard_bounds(200, r, 3) for r in range(3)] == [(0, 128), (128, 192), (192, 200)]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rank0.npz")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    assert got["total"] == R_TOTAL and got["worst"] == 1.0
    whole = synthetic.make_batch(R_TOTAL, E, n_steps=STEPS, ego_kind=3, extent=20.0)
    assert np.array_equal(got["rows"], _rows_for(whole, oracle))  # same bits as one unsharded run


from standin_engine import StandInEngine as _StandInEngine  # noqa: E402


def _bench_worker(rank, world, port, out_path):
    import json
    import sys

    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench
    from scenario_gym_amd import distributed as D

    D.init("gloo")  # bench.main() finds the process group initialised
    made = []

    def make_engine(R, first):
        made.append((R, first))
        return _StandInEngine(R, first, 6, 40)

    line = bench.main(["--gpus", str(world), "--scenarios", "128", "--entities", "6", "--sim-steps", "40", "--steps", "2",
                       "--warmup", "1", "--scaling", "strong", "--no-cpu-baseline"], make_engine=make_engine)
    assert made == [(64, rank * 64), (128, rank * 128)]  # the strong shape first (it is `value`), then the weak one
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump(line, f)
    else:
        assert line is None
    import torch.distributed as dist

    dist.barrier()
    dist.destroy_process_group()


def test_bench_dispatch_and_collection_strong_scaling(tmp_path):
    """bench.py's own dispatch / timed passes / gather / aggregation with --scaling strong on two gloo ranks: `value` is
    the 128 scenarios split 64 + 64, the weak shape (128 per rank) is reported beside it, ranks / backend / per-rank
    values are in the line."""
    import json

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "line.json")
    mp.spawn(_bench_worker, args=(2, port, out), nprocs=2, join=True)
    line = json.load(open(out))
    assert line["scaling"] == "strong" and line["ranks"] == 2 and line["n_gpus"] == 2 and line["backend"] == "gloo"
    assert line["config"]["scenarios_per_gpu"] == 64 and line["weak"]["scenarios_per_gpu"] == 128
    assert len(line["per_rank_value"]) == 2 and len(line["weak"]["per_rank_value"]) == 2
    ent_steps = 2 * 64 * 6 * 40 * 2  # ranks x scenarios x entities x T x timed passes
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 * 2 - ent_steps) < 1e-6 * ent_steps
    # (no counted flops for this toy shape: the contract's HBM figure stands in; the stand-in engine has no schedule to report,
    # and the line says which engine produced it)
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["binding"] == "valu_issue" and line["roofline"]["traffic"] is None
    assert line["roofline"]["schedule"] is None and "degraded" not in line and line["engine"] == "injected stand-in (tests)"
    assert line["roofline"]["entity_steps_per_launch"] == 64 * 6 * 40 / 2


def test_bench_py_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` the way the driver invokes it (no launcher, WORLD_SIZE unset): bench.py starts the two
    ranks itself (torch.distributed.run as a child, gloo here), relays rank 0's line: ranks 2, n_gpus 2.  With more than one
    rank `value` is BASELINE.json's metric as written -- the batch SPLIT over the ranks (scaling "strong", configs[3]) -- and the
    weak shape (the whole batch per rank) rides beside it."""
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scenarios", "128", "--entities", "6",
                          "--sim-steps", "40", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                          "--engine-factory", "tests.standin_engine:make"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["ranks"] == 2 and line["n_gpus"] == 2 and line["backend"] == "gloo" and line["scaling"] == "strong"
    assert line["config"]["scenarios_per_gpu"] == 128 // 2 and line["weak"]["scenarios_per_gpu"] == 128
    assert len(line["per_rank_value"]) == 2 and len(line["weak"]["per_rank_value"]) == 2
    assert line["engine"] == "tests.standin_engine:make"
    ent_steps = 2 * 64 * 6 * 40 * 2
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 * 2 - ent_steps) < 1e-6 * ent_steps
