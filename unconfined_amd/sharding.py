"""Multi-GPU layout of a sweep, one process per GPU.

Every (t,r) point is independent, so the path shards with no data-path collective.  The shard axis is the reference's
own serial loop nest (``do i = 1,nt / do k = 1,nr``, reference driver.f90:100,113): the flattened index ``i*nr + k`` is
cut into ``world`` contiguous blocks of whole time rows -- ``ucf_shard_rows`` of the C ABI, the one partition rule that
the library's own multi-GPU entry points (``ucf_drawdown_grid_shard_device``, ``ucf_drawdown_grid_multi``), ``bench.py``
and the tests all use.  Every rank writes its rows at their place in full-size result arrays padded to ``world * B``
rows, B = ceil(nt / world), and ONE in-place all-gather per array (RCCL over xGMI on GPUs, gloo in the CPU tests)
completes them on every rank -- the only exchange of the path.
"""
from __future__ import annotations

from typing import Tuple

from .engine import shard_rows  # noqa: F401  (re-exported: the partition rule)


def rows_per_shard(nt: int, world: int) -> int:
    """B = ceil(nt / world): what ucf_shard_rows gives every shard but the last ones"""
    return (nt + world - 1) // world


def padded_rows(nt: int, world: int) -> int:
    """rows of the full-size arrays that an in-place all-gather of equal slices needs"""
    return rows_per_shard(nt, world) * world


def shard_slice(nt: int, row_elems: int, world: int, rank: int) -> Tuple[int, int]:
    """[start, stop) of shard `rank` in a flat array of padded_rows(nt, world) * row_elems elements"""
    B = rows_per_shard(nt, world)
    return rank * B * row_elems, (rank + 1) * B * row_elems


def allgather_rows_(full, nt: int, row_elems: int, world: int, rank: int, group=None):
    """in-place all-gather of a flat [padded_rows * row_elems] tensor whose slice `rank` this rank has filled
    (rows beyond nt in the last shards are padding and carry whatever the buffer held)"""
    import torch.distributed as dist
    if world == 1:
        return full
    a, b = shard_slice(nt, row_elems, world, rank)
    if full.numel() != padded_rows(nt, world) * row_elems:
        raise ValueError("full-size array must hold padded_rows(nt, world) * row_elems elements")
    mine = full[a:b]
    if dist.get_backend(group) != "nccl":
        mine = mine.clone()            # gloo does not take an input that aliases the output
    dist.all_gather_into_tensor(full, mine, group=group)
    return full
