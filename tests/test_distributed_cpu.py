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
