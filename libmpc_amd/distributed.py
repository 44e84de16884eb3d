"""Multi-GPU sharding of a batch of MPC instances: one process per GPU.

Instances are independent (the reference solves one controller at a time and has no
coupling between controllers), so the batch shards as contiguous slices with no
collective on the data path; the only exchange is one all-gather of the optimal
controls u* (B x nu doubles) after the solve -- RCCL over xGMI when the process group
is "nccl", gloo in the CPU tests."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int):
    """Contiguous slice [lo, hi) of `total` instances owned by `rank`; the first
    (total % world) ranks get one extra instance."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def allgather_controls(cmd_local: torch.Tensor, total: int | None = None, group=None, force: bool = False) -> torch.Tensor:
    """All-gather the per-rank optimal controls into [total, nu] on every rank.

    Equal shards use one all_gather_into_tensor (a single RCCL ring collective);
    ragged shards are padded to the largest shard and trimmed afterwards."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return cmd_local           # force: run the collective even on a single rank (exercises the RCCL path)
    world = dist.get_world_size(group)
    n_local, nu = cmd_local.shape
    if total is None or total == n_local * world:
        out = torch.empty((world * n_local, nu), dtype=cmd_local.dtype, device=cmd_local.device)
        dist.all_gather_into_tensor(out, cmd_local.contiguous(), group=group)
        return out
    sizes = [shard_range(total, r, world) for r in range(world)]
    nmax = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((nmax, nu), dtype=cmd_local.dtype, device=cmd_local.device)
    pad[:n_local] = cmd_local
    out = torch.empty((world * nmax, nu), dtype=cmd_local.dtype, device=cmd_local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    parts = [out[r * nmax: r * nmax + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(parts, dim=0)
