"""Replica sharding across the GPUs of one node (one process per GPU, torch.distributed).

Scenarios never interact (the reference runs them one after another: scenario_gym.py:24-27,
manager.py:272-282), so the batch shards along the replica axis with NO collective on the step path.
RCCL ("nccl" backend on ROCm) -- or gloo on CPU in the tests -- is used for exactly two things:
  dispatch    rank 0 broadcasts the run configuration; every rank generates / receives its shard
  collection  per-replica metric rows are gathered on rank 0
"""
import os
from typing import Optional, Tuple

import numpy as np

from .synthetic import CHUNK


def init(backend: Optional[str] = None):
    """(rank, world, local_rank, dist-or-None) from the torchrun environment."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and not os.environ.get("SGYM_FORCE_DIST"):  # SGYM_FORCE_DIST=1: exercise the collectives with one rank
        return rank, world, local_rank, None
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # SGYM_DIST_BACKEND=gloo + SGYM_DIST_ONE_DEVICE=1: several ranks on ONE GPU (RCCL refuses two ranks on a device; gloo moves
    # the few dispatch / collection bytes over the host) -- how the live N > 1 path is exercised on a one-GPU box
    if os.environ.get("SGYM_DIST_ONE_DEVICE"):
        local_rank = 0
    if not dist.is_initialized():
        backend = backend or os.environ.get("SGYM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, **kw)
    return rank, world, local_rank, dist


def _device(dist):
    import torch

    if dist is not None and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def dispatch_config(values, dist, src: int = 0):
    """Broadcast a flat list of int64 run parameters from `src` (replica dispatch)."""
    import torch

    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=_device(dist))
    if dist is not None:
        dist.broadcast(t, src=src)
    return [int(v) for v in t.tolist()]


def shard_bounds(n_total: int, rank: int, world: int, chunk: int = CHUNK) -> Tuple[int, int]:
    """Contiguous [lo, hi) of scenarios for `rank`; shards start on generator-chunk boundaries."""
    n_chunks = (n_total + chunk - 1) // chunk
    per, extra = divmod(n_chunks, world)
    c_lo = rank * per + min(rank, extra)
    c_hi = c_lo + per + (1 if rank < extra else 0)
    return min(c_lo * chunk, n_total), min(c_hi * chunk, n_total)


def gather_rows(rows: np.ndarray, dist, dst: int = 0) -> Optional[np.ndarray]:
    """Gather [R_rank, M] float64 rows from every rank on `dst` (metric collection).
    Shards may differ in length: sizes are exchanged first, payloads are padded to the maximum."""
    import torch

    rows = np.ascontiguousarray(rows, np.float64)
    if dist is None:
        return rows
    dev = _device(dist)
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    pad = np.zeros((max(sizes), rows.shape[1]))
    pad[: rows.shape[0]] = rows
    t = torch.from_numpy(pad).to(dev)
    out = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
    dist.gather(t, out, dst=dst)
    if rank != dst:
        return None
    return np.concatenate([o.cpu().numpy()[:s] for o, s in zip(out, sizes)], axis=0)


def max_over_ranks(x: float, dist) -> float:
    import torch

    if dist is None:
        return float(x)
    t = torch.tensor([x], dtype=torch.float64, device=_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x: float, dist) -> float:
    import torch

    if dist is None:
        return float(x)
    t = torch.tensor([x], dtype=torch.float64, device=_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
