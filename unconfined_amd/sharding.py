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


def library_communicator(world: int, rank: int, device, carry_id=None, timeout: float = 120.0):
    """The library's own RCCL communicator for ``ucf_drawdown_grid_allgather`` (``ucf_comm_unique_id`` on rank 0,
    ``ucf_comm_create`` everywhere), made so that no rank can be left waiting for another:

    * rank 0 draws the 128-byte id; ``carry_id(buf129)`` -- the host's transport, e.g. a ``torch.distributed`` broadcast of
      a uint8 tensor from rank 0 -- carries it to the others TOGETHER with a validity byte, and is called on every rank
      whether or not the draw succeeded (a rank that skipped it would leave the others inside the broadcast);
    * ``ncclCommInitRank`` is a collective: it runs on a helper thread (with the rank's device current -- the HIP device is
      per thread) and is given ``timeout`` seconds; a rank that is still inside after that reports failure and leaves the
      thread behind rather than hang the job.

    Returns ``(comm, error)``: the communicator handle, or ``0`` and the reason.  The caller still has to agree over its own
    transport that EVERY rank got one (all or none) before the first collective."""
    import threading
    import torch
    from . import engine
    buf = torch.zeros(129, dtype=torch.uint8)
    err = None
    if rank == 0:
        try:
            buf[:128] = torch.frombuffer(bytearray(engine.comm_unique_id()), dtype=torch.uint8)
            buf[128] = 1
        except Exception as exc:          # no RCCL in the process and none to load
            err = f"ucf_comm_unique_id: {exc}"
    if carry_id is not None and world > 1:
        buf = carry_id(buf)
    if int(buf[128]) != 1:
        return 0, err or "rank 0 could not draw a communicator id"
    uid = bytes(buf[:128].cpu().numpy().tobytes())
    box = {}

    def make():
        try:
            if device is not None:
                torch.cuda.set_device(device)
            box["comm"] = engine.comm_create(uid, world, rank)
        except Exception as exc:
            box["err"] = f"ucf_comm_create: {exc}"

    th = threading.Thread(target=make, name="ucf-comm-create", daemon=True)
    th.start()
    th.join(timeout)
    if th.is_alive():
        return 0, f"ucf_comm_create did not return within {timeout:.0f} s"
    if "comm" in box:
        return box["comm"], None
    return 0, box.get("err", "ucf_comm_create failed")
