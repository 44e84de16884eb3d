"""Multi-GPU sharding of a batch of MPC instances: one process per GPU.

Instances are independent (the reference solves one controller at a time and has no
coupling between controllers), so the batch shards as contiguous slices with no
collective on the data path; the only exchange is one all-gather of the optimal
controls u* (B x nu doubles) after the solve.

On the GPU that all-gather goes through the C ABI (`mpcx_comm_*` / `mpcx_allgather_u`,
include/mpcx.h): RCCL's ncclAllGather over xGMI, enqueued on the stream the solve kernels
were launched on -- the same entry point a C++ host uses.  `ControlGather` wraps it; the
only thing it borrows from torch.distributed is the rendezvous (the 128-byte RCCL id
travels from rank 0 to the others through the process group's store).  CPU tensors (the
gloo tests) take torch's own all_gather."""
from __future__ import annotations

import ctypes as C

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int):
    """Contiguous slice [lo, hi) of `total` instances owned by `rank`; the first
    (total % world) ranks get one extra instance."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


class ControlGather:
    """One RCCL communicator per process, behind the C ABI.  `rank`/`world` default to the
    initialised torch.distributed group, which is then also used to ship the RCCL id; a
    single-rank communicator needs no process group at all."""

    def __init__(self, device: int, rank: int | None = None, world: int | None = None, group=None):
        from . import _capi
        self._lib = _capi.lib()
        self._check = _capi.check
        if rank is None or world is None:
            if dist.is_initialized():
                rank, world = dist.get_rank(group), dist.get_world_size(group)
            else:
                rank, world = 0, 1
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        ident = (C.c_ubyte * 128)()
        if self.rank == 0:
            self._check(self._lib.mpcx_comm_get_unique_id(ident))
        if self.world > 1:
            box = [bytes(ident)]
            # rendezvous only: 128 bytes through the store; the source is the group's rank 0 as a GLOBAL rank (sub-groups)
            src = dist.get_global_rank(group, 0) if (group is not None and dist.is_initialized()) else 0
            dist.broadcast_object_list(box, src=src, group=group)
            ident = (C.c_ubyte * 128).from_buffer_copy(box[0])
        self._h = C.c_void_p()
        self._check(self._lib.mpcx_comm_create(self.device, self.rank, self.world, ident, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.mpcx_comm_destroy(self._h)
            self._h = None

    def allgather(self, cmd_local: torch.Tensor, out: torch.Tensor | None = None, stream=None) -> torch.Tensor:
        """cmd_local [n, nu] (cuda, fp64, the same n on every rank) -> [world * n, nu], enqueued on `stream`
        (default: torch's current stream on the communicator's device)."""
        assert cmd_local.is_cuda and cmd_local.dtype == torch.float64 and cmd_local.is_contiguous()
        n, nu = cmd_local.shape
        if out is None:
            out = torch.empty((self.world * n, nu), dtype=torch.float64, device=cmd_local.device)
        s = torch.cuda.current_stream(cmd_local.device).cuda_stream if stream is None else stream
        self._check(self._lib.mpcx_allgather_u(self._h, cmd_local.data_ptr(), int(n), int(nu), out.data_ptr(), C.c_void_p(s)))
        return out


def allgather_controls(cmd_local: torch.Tensor, total: int | None = None, group=None, force: bool = False,
                       gather: ControlGather | None = None) -> torch.Tensor:
    """All-gather the per-rank optimal controls into [total, nu] on every rank.

    Equal shards use one all-gather (RCCL through `gather` for CUDA tensors, torch's
    collective otherwise); ragged shards are padded to the largest shard and trimmed
    afterwards."""
    if gather is None and (not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force)):
        return cmd_local           # force: run the collective even on a single rank (exercises the path)
    world = gather.world if gather is not None else dist.get_world_size(group)
    n_local, nu = cmd_local.shape

    def collect(block):
        if gather is not None and block.is_cuda:
            return gather.allgather(block.contiguous())
        out = torch.empty((world * block.shape[0], nu), dtype=block.dtype, device=block.device)
        dist.all_gather_into_tensor(out, block.contiguous(), group=group)
        return out

    if total is None or total == n_local * world:
        return collect(cmd_local)
    sizes = [shard_range(total, r, world) for r in range(world)]
    nmax = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((nmax, nu), dtype=cmd_local.dtype, device=cmd_local.device)
    pad[:n_local] = cmd_local
    out = collect(pad)
    parts = [out[r * nmax: r * nmax + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(parts, dim=0)


class OverlappedGather:
    """Double buffering of the one exchange of the data path: step k's solve writes its optimal controls into buffer k % 2 on the solve
    stream while step k-1's all-gather (buffer (k-1) % 2) is still travelling on a second stream.  At a 50 us step an all-gather issued in
    series on the solve stream (latency-bound: tens of microseconds over xGMI) would be a large part of every step; behind an event on its
    own stream it costs the solve stream nothing.

    Ordering, per buffer i: solve(k) waits for the gather of step k-2 (the last reader of buffer i) -> `done[i]` -> gather(k) on the
    gather stream -> `free[i]`.  `gathered(k)` is the [world * n, nu] result of step k, valid until step k+2 is launched.

    CUDA tensors: two HIP streams and events, the collective through `ControlGather` (RCCL behind the C ABI).  CPU tensors (the gloo
    tests): the same rotation with torch's collective and no streams -- what is tested there is that the results of step k are what is
    gathered as step k, and that they survive the launch of step k+1."""

    def __init__(self, n: int, nu: int, device, gather: ControlGather | None = None, group=None, solve_stream=None, world: int | None = None):
        self.gather, self.group = gather, group
        self.world = gather.world if gather is not None else (world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1))
        dev = torch.device(device)
        self.cuda = dev.type == "cuda"
        self.cmd = [torch.zeros((n, nu), dtype=torch.float64, device=dev) for _ in range(2)]
        self.all = [torch.zeros((self.world * n, nu), dtype=torch.float64, device=dev) for _ in range(2)]
        self.k_of = [-1, -1]                                   # which step's results each buffer holds
        if self.cuda:
            self.solve_stream = solve_stream if solve_stream is not None else torch.cuda.current_stream(dev)
            self.gather_stream = torch.cuda.Stream(device=dev)
            self.done = [torch.cuda.Event() for _ in range(2)]
            self.free = [torch.cuda.Event() for _ in range(2)]
            for e in self.free:
                e.record(self.solve_stream)

    def step(self, k: int, launch):
        """launch(i, stream) enqueues (CUDA) or performs (CPU) the solve of step k with its controls going to `self.cmd[i]`"""
        i = k % 2
        if self.cuda:
            self.solve_stream.wait_event(self.free[i])         # buffer i's previous all-gather has read it
            launch(i, self.solve_stream)
            self.done[i].record(self.solve_stream)
            self.gather_stream.wait_event(self.done[i])
            if self.gather is not None:
                self.gather.allgather(self.cmd[i], out=self.all[i], stream=self.gather_stream.cuda_stream)
            else:
                with torch.cuda.stream(self.gather_stream):
                    self.all[i].copy_(self.cmd[i].repeat(self.world, 1) if self.world > 1 else self.cmd[i])
            self.free[i].record(self.gather_stream)
        else:
            launch(i, None)
            if dist.is_initialized() and self.world > 1:
                dist.all_gather_into_tensor(self.all[i], self.cmd[i].contiguous(), group=self.group)
            else:
                self.all[i].copy_(self.cmd[i])
        self.k_of[i] = k

    def gathered(self, k: int) -> torch.Tensor:
        i = k % 2
        if self.k_of[i] != k:
            raise RuntimeError("step %d is no longer (or not yet) in its buffer: it holds step %d" % (k, self.k_of[i]))
        if self.cuda:
            self.free[i].synchronize()
        return self.all[i]

    def finish(self):
        if self.cuda:
            self.gather_stream.synchronize()


class GraphedOverlap:
    """`OverlappedGather` without its host calls: {solve of step k on the capture stream || all-gather of step k-1 on a second stream}
    captured once per buffer parity as ONE HIP graph (fork at the graph's root, join at its end) and replayed with one launch per step.
    The eager form orders the two streams with five host calls per step (two waits, two records, a second-stream launch), which at a 50 us
    step is what the loop then waits for; a graph launch is one.  RCCL (>= 2.9) collectives can be captured; so can the solve (it is what
    `mpcx_lmpc_graph_create` captures).  After graph k: `cmd[k % 2]` holds step k's controls, `all[(k - 1) % 2]` the gathered controls of
    step k - 1; `flush()` gathers the last step's.  One instance per (controller, pair of batch descriptors): the descriptors' pointers are
    baked into the graphs, what they point to may change between launches."""

    def __init__(self, og: OverlappedGather, launch, warmup: int = 2):
        assert og.cuda and og.gather is not None
        self.og, self.launch = og, launch
        dev = og.cmd[0].device
        # eager rounds first: allocations, function attributes, RCCL's lazy set-up must not happen inside a capture
        for k in range(2 * warmup):
            og.step(k, launch)
        og.finish()
        torch.cuda.synchronize(dev)
        self.stream = torch.cuda.Stream(device=dev)
        self.graphs = []
        for i in (0, 1):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self.stream):
                cur = torch.cuda.current_stream(dev)
                og.gather_stream.wait_stream(cur)                  # fork at the root: nothing of this step is enqueued yet
                launch(i, cur)                                     # solve of step k -> cmd[i]
                og.gather.allgather(og.cmd[1 - i], out=og.all[1 - i], stream=og.gather_stream.cuda_stream)      # step k-1's controls
                cur.wait_stream(og.gather_stream)                  # join
            self.graphs.append(g)
        self.k = 0

    def step(self):
        with torch.cuda.stream(self.stream):
            self.graphs[self.k % 2].replay()
        self.k += 1

    def flush(self):
        """the all-gather of the last step (its graph has not been launched), in series on the graphs' stream"""
        i = (self.k - 1) % 2
        self.og.gather.allgather(self.og.cmd[i], out=self.og.all[i], stream=self.stream.cuda_stream)
        self.stream.synchronize()
        return self.og.all[i]
