"""N > 1 path: the partition rule of the C ABI (ucf_shard_rows) and the in-place all-gather that bench.py and the
library's multi-GPU entry points use, driven by world_size-2/3 gloo processes on the CPU; on the GPU box
additionally bench.py --gpus 2 starting its own ranks (both on the one GPU, gloo) through exactly that code."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from unconfined_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_rows_cover_the_sweep():
    """contiguous blocks of whole time rows (driver.f90:100), B = ceil(nt / world) each, last ones short or empty"""
    for nt in (0, 1, 7, 64, 1024, 1025, 4096):
        for world in (1, 2, 3, 8):
            blocks = [sharding.shard_rows(nt, world, g) for g in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == nt
            for a, b in zip(blocks, blocks[1:]):
                assert a[1] == b[0]
            B = sharding.rows_per_shard(nt, world)
            assert all(hi - lo <= B for lo, hi in blocks)
            assert all(hi - lo == B for lo, hi in blocks if hi < nt)
            assert sharding.padded_rows(nt, world) == B * world >= nt
            for g, (lo, hi) in enumerate(blocks):
                a, b = sharding.shard_slice(nt, 5, world, g)
                assert a == g * B * 5 and b - a == B * 5 and (lo == hi or a == lo * 5)
    from unconfined_amd.lib import UcfError
    with pytest.raises(UcfError):
        sharding.shard_rows(10, 2, 2)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _value(i, k, which):          # stand-in for (h, dh) of point (row i, column k): the CPU has no drawdown path
    return (i * 1000.0 + k) * (2.0 if which == 0 else -3.0) + which


def _worker(rank, world, port, nt, row, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.shard_rows(nt, world, rank)
    ok = True
    for which in (0, 1):
        full = torch.full((sharding.padded_rows(nt, world) * row,), float("nan"), dtype=torch.float64)
        ii, kk = np.meshgrid(np.arange(lo, hi), np.arange(row), indexing="ij")
        full[lo * row: hi * row] = torch.from_numpy(_value(ii, kk, which).ravel())     # this rank's rows, in place
        sharding.allgather_rows_(full, nt, row, world, rank)
        ii, kk = np.meshgrid(np.arange(nt), np.arange(row), indexing="ij")
        ok = ok and bool(torch.equal(full[: nt * row], torch.from_numpy(_value(ii, kk, which).ravel())))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nt", [(2, 10), (2, 1025), (3, 64)])
def test_inplace_allgather_gloo(world, nt):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nt, 6, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


def test_bench_refuses_a_wrong_world():
    """--gpus must equal the ranks that exist: a rank whose WORLD_SIZE differs exits non-zero before any GPU work"""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 5 and "refusing" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_bench_starts_its_own_ranks(scaling):
    """`python bench.py --gpus 2` with no launcher around it: two ranks (rehearsal: both on cuda:0, gloo), the fixed
    sweep sharded by rows / one sweep per rank, gathered and identical on both ranks"""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, UCF_BENCH_ONE_DEVICE="1", UCF_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--nt", "256",
                        "--nr", "24", "--scaling", scaling], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == scaling
    assert line["config"]["results_finite_and_gather_consistent"] is True
    assert line["config"]["points_per_step"] == (256 * 24 if scaling == "strong" else 256 * 24 * 2)
    assert line["other_scaling"]["scaling"] != scaling and line["other_scaling"]["value"] > 0


@pytest.mark.gpu
def test_bench_fails_without_enough_gpus():
    """one GPU on the box, --gpus 2 with the real backend: the missing rank makes the whole run fail"""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "UCF_BENCH_ONE_DEVICE", "UCF_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--nt", "64", "--nr", "4"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())
