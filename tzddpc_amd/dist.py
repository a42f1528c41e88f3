"""Multi-GPU plumbing: trajectories are independent units, so the batch is sharded across ranks with no
data-path collective; the only exchange is one all-gather of per-trajectory (cost, final state) -- RCCL over
xGMI when the backend is "nccl", gloo in the CPU tests.  One process per GPU (torch.distributed)."""
from __future__ import annotations

from typing import Tuple

import numpy as np


def shard_range(total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of `total` trajectories owned by `rank` (sizes differ by at most one)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank outside world")
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_results(local, total: int):
    """All-gather rows of a (local_B, k) tensor from every rank into (total, k), in shard order.

    Shards may differ by one row: they are padded to the largest shard for the collective and trimmed after.
    """
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return local
    ws, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(total, ws, r) for r in range(ws)]
    maxb = max(hi - lo for lo, hi in sizes)
    out = torch.empty((ws * maxb, local.shape[1]), dtype=local.dtype, device=local.device)
    if total == ws * maxb:                     # equal shards: one collective straight from the caller's buffer, no padding, no trimming
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    pad = torch.zeros((maxb, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    dist.all_gather_into_tensor(out, pad)
    parts = [out[r * maxb: r * maxb + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(parts, dim=0)


def vertex_noise(W_vertices: np.ndarray, first_trajectory: int, count: int, steps: int, seed_base: int = 1000) -> np.ndarray:
    """Per-trajectory process noise: uniformly random vertex of W, generator PCG64(seed_base + global index)
    (SURVEY.md section 8d config 2), so a trajectory's noise does not depend on how the batch is sharded."""
    out = np.empty((count, steps, W_vertices.shape[1]))
    for i in range(count):
        rng = np.random.Generator(np.random.PCG64(seed_base + first_trajectory + i))
        out[i] = W_vertices[rng.integers(len(W_vertices), size=steps)]
    return out


def sync_calibration(values, device=None):
    """Build-time calibration chosen by every rank on its own box (warm-start shift policy, push gain / cap, complementarity
    factor: short closed loops whose factorisation counts can tie differently on two devices): rank 0's choice is broadcast and
    returned for every rank to adopt, together with whether all ranks had chosen the same on their own.
    values: sequence of floats (inf allowed).  -> (rank 0's values as a list of floats, identical_across_ranks: bool)."""
    import torch
    import torch.distributed as dist
    mine = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if not dist.is_initialized():
        return [float(v) for v in mine.cpu()], True
    lead = mine.clone()
    dist.broadcast(lead, src=0)
    same = torch.tensor([1.0 if torch.equal(lead, mine) else 0.0], dtype=torch.float64, device=device)
    dist.all_reduce(same, op=dist.ReduceOp.MIN)
    return [float(v) for v in lead.cpu()], bool(same.item() == 1.0)
