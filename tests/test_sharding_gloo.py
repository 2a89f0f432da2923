"""N > 1 path on CPU: world_size-2 gloo processes partition a sweep and all-gather results."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from unconfined_amd.sharding import block_partition, gather_blocks


def test_block_partition_covers_everything():
    for npts in (1, 7, 64, 262144, 262145):
        for world in (1, 2, 3, 8):
            blocks = [block_partition(npts, world, g) for g in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == npts
            for a, b in zip(blocks, blocks[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, npts, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = block_partition(npts, world, rank)
    idx = torch.arange(lo, hi, dtype=torch.float64)
    local = torch.stack([idx * 2.0, idx * 3.0 + 1.0], dim=1)       # stand-in for (h, dh) of each owned point
    full = gather_blocks(local, npts, world, rank)
    ok = bool(torch.equal(full[:, 0], torch.arange(npts, dtype=torch.float64) * 2.0)) and \
        bool(torch.equal(full[:, 1], torch.arange(npts, dtype=torch.float64) * 3.0 + 1.0))
    q.put((rank, ok, tuple(full.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("npts", [10, 1025])
def test_gather_blocks_world2_gloo(npts):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, npts, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert all(shape == (npts, 2) for _, _, shape in res)
