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


def _overlap_worker(rank, world, port, n, nu, steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from libmpc_amd.distributed import OverlappedGather
    og = OverlappedGather(n, nu, "cpu")

    def value(r, k):                                             # what rank r's solve of step k "returns"
        return (1000.0 * k + 10.0 * r) + torch.arange(n * nu, dtype=torch.float64).reshape(n, nu)

    ok = True
    for k in range(steps):
        og.step(k, lambda i, stream: og.cmd[i].copy_(value(rank, k)))
        want = torch.cat([value(r, k) for r in range(world)], dim=0)
        ok &= torch.equal(og.gathered(k), want)                  # the results of step k are what was gathered as step k ...
        if k >= 1:
            prev = torch.cat([value(r, k - 1) for r in range(world)], dim=0)
            ok &= torch.equal(og.gathered(k - 1), prev)          # ... and step k-1's survive the launch of step k (two buffers)
        if k >= 2:
            try:
                og.gathered(k - 2)
                ok = False                                       # its buffer holds step k by now: refused, not returned stale
            except RuntimeError:
                pass
    og.finish()
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_double_buffered_gather_keeps_step_order_gloo():
    """bench.py --gpus N overlaps step k's all-gather with step k+1's solve (libmpc_amd.distributed.OverlappedGather); the rotation of the
    two buffers on a world-size-2 gloo group"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, 6, 4, 5, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res
