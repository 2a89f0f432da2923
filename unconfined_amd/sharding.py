"""Multi-GPU layout of a sweep: every (t,r) point is independent,
so the flattened point index is block-partitioned over ranks, each rank runs the
fused kernel on its block with no data-path collective, and one all-gather (RCCL
over xGMI on GPUs, gloo in the CPU tests) reassembles the [npts, 2*nz] result."""
from __future__ import annotations

from typing import Tuple


def block_partition(npts: int, world: int, rank: int) -> Tuple[int, int]:
    """contiguous block [lo, hi) of rank `rank`: idx = i_t*nr + i_r, rank g owns
    [g*P/G, (g+1)*P/G) (integer arithmetic; blocks differ by at most one point)"""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    lo = (npts * rank) // world
    hi = (npts * (rank + 1)) // world
    return lo, hi


def gather_blocks(local, npts: int, world: int, rank: int, group=None):
    """all-gather variable-size contiguous blocks of a [n_local, C] tensor into [npts, C].
    Blocks are padded to the largest block so that one all_gather_into_tensor suffices."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local
    sizes = [block_partition(npts, world, g)[1] - block_partition(npts, world, g)[0] for g in range(world)]
    mx = max(sizes)
    C = local.shape[1]
    pad = torch.zeros(mx, C, dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty(world * mx, C, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    out = out.view(world, mx, C)
    return torch.cat([out[g, : sizes[g]] for g in range(world)], dim=0)
