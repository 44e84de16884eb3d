"""The N>1 path on CPU: world_size-2 gloo process group, sharding + all-gather of u*."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from libmpc_amd.distributed import allgather_controls, shard_range


def test_shard_range_covers_batch():
    for total in (0, 1, 7, 4096, 262144):
        for world in (1, 2, 3, 8):
            edges = [shard_range(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[r][1] == edges[r + 1][0] for r in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, total, nu, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(total, rank, world)
    full = torch.arange(total * nu, dtype=torch.float64).reshape(total, nu)
    out = allgather_controls(full[lo:hi].clone(), total=total)
    ok = torch.equal(out, full)
    q.put((rank, bool(ok), tuple(out.shape)))
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 9])
def test_allgather_controls_gloo(total):
    world, nu = 2, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + total
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, nu, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert all(shape == (total, nu) for _, _, shape in res)
