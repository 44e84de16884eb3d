"""bench.py -- MPC solves/sec of the batched solve path on MI355X.

Default workload: BASELINE.json's headline, the reference's quadrotor LMPC (N=20), batch 4096 per GPU.  One "step" = one
pass of the hot path (batched LOptimizer::run / NLOptimizer::run) over one batch of synthetic instances already resident
in HBM.  N>1: one process per GPU; each rank solves its own shard, then one RCCL all-gather of u* over xGMI issued through
the C ABI (mpcx_allgather_u) on the solve stream; weak scaling.  Prints ONE JSON line on rank 0.

  python bench.py                               # config 2: quadrotor LMPC N=20, 4096 instances, 1 GPU
  python bench.py --gpus 8                      # spawns 8 ranks itself (torch.distributed.run); the driver may also
                                                # launch the ranks, then RANK/WORLD_SIZE come from the environment
  python bench.py --config 4 --gpus 8           # quadrotor N=50, 32768 per GPU (262144 on 8), 8 MiB all-gather
  python bench.py --workload ugv|osc8|vanderpol|osc6   # the NLMPC configs (3, 5, 1) under the same contract
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_FP64_TFLOPS = 78.6      # MI355X FP64 vector == FP64 matrix peak; see DESIGN.md
PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s

CONFIGS = {1: ("vanderpol", 4096), 2: ("lmpc", 4096), 3: ("ugv", 4096), 4: ("lmpc50", 32768), 5: ("osc8", 1024)}


def kernel_source_hash():
    """what the committed PMC profiles are stamped with: a digest of every source the device code is built from"""
    h = hashlib.sha256()
    files = []
    for d, pat in (("libmpc_amd/csrc", (".hip", ".hpp", ".cpp")), ("include/mpcx", (".hpp",))):
        for f in sorted(os.listdir(os.path.join(ROOT, d))):
            if f.endswith(pat):
                files.append(os.path.join(ROOT, d, f))
    for f in files:
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _usable_cores():
    """host cores this process may actually use: the affinity mask, cut down to the cgroup CPU quota if there is one"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, int(q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def _cpu_worker(job):
    """one host core's share of the CPU baseline: the C oracle on a slice of the batch (fork()ed worker)"""
    ph, x0, u0, yref, budget = job
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import quadrotor_oracle
    o = quadrotor_oracle(ph)
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget:          # the slice again and again until the time budget is used
        o.solve_batch_constref(x0, u0, yref)
        done += int(x0.shape[0])
    return done, time.perf_counter() - t0


def _hetero_cpu_worker(job):
    """heterogeneous LMPC CPU baseline, one host core's share: instance k on the C oracle configured as controller quadrotor_variant(k) -- one
    controller object per problem, configured once outside the timed loop (LMPC.hpp:751), the set-up inside every solve as LOptimizer::run
    pays it -- until the budget is used"""
    ph, first, x0, u0, yref, budget = job
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import OracleFrontEnd
    from libmpc_amd.workloads import quadrotor_variant
    from oracle.lmpc_oracle import default_params
    ctrls = []
    for t in range(x0.shape[0]):
        f = OracleFrontEnd(12, 4, 4, 12, ph, ph)
        quadrotor_variant(first + t, ph, into=f)
        f.o.params = default_params(maximum_iteration=250)
        ctrls.append(f.o)
    done, t0 = 0, time.perf_counter()
    while True:
        for t, o in enumerate(ctrls):
            o.solve_batch_constref(x0[t:t + 1], u0[t:t + 1], yref[t:t + 1])
            done += 1
            if time.perf_counter() - t0 > budget:
                return done, time.perf_counter() - t0


def _nl_cpu_worker(job):
    """NLMPC CPU baseline, one host core's share: the reference's callbacks in C (oracle/nlmpc_callbacks.c) driving scipy's SLSQP
    (Kraft's compiled code, what NLopt's LD_SLSQP translates) on instances of the batch, until the time budget is used"""
    name, x0, u0, budget = job
    from oracle import nlmpc_c
    m = nlmpc_c.make(name)
    done, t0 = 0, time.perf_counter()
    while True:
        for i in range(x0.shape[0]):
            m.solve(x0[i], u0[i], max_iter=150, hard=(name != "ugv"))
            done += 1
            if time.perf_counter() - t0 > budget:
                return done, time.perf_counter() - t0


def _sq_counters(kernel, tag):
    """unit-busy fractions of one kernel from the committed hardware-counter summary of this command (tools/profile_sq.sh):
    VALU / MFMA / LDS / scalar busy cycles over the cycles its wavefronts were resident (all SQ counters count 4-cycle steps
    per wavefront), L2 hit rate.  Same staleness rule as the traffic figure."""
    import glob
    cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sq_%s.json" % tag)))
    if not cand:
        return None
    pm = json.load(open(cand[-1]))
    if pm.get("kernel_source_hash") != kernel_source_hash():
        return {"source": os.path.basename(cand[-1]), "note": "counter summary is stale: kernel sources changed since it was measured"}
    stat = "max" if kernel.startswith("nlmpc_sqp_wg") else "mean"      # (two launches per solve there, the second all but empty: see _traffic)
    g = lambda c: pm.get(c, {}).get(kernel, {}).get(stat)
    wc = g("SQ_WAVE_CYCLES")
    if not wc:
        return {"source": os.path.basename(cand[-1]), "note": "kernel %s not in the counter summary" % kernel}
    frac = lambda c: (g(c) / wc) if g(c) is not None else None
    hit, miss = g("TCC_HIT"), g("TCC_MISS")
    return {"source": os.path.basename(cand[-1]), "valu_busy_per_wave": frac("SQ_ACTIVE_INST_VALU"), "mfma_busy_per_wave": frac("SQ_VALU_MFMA_BUSY_CYCLES"),
            "lds_busy_per_wave": frac("SQ_ACTIVE_INST_LDS"), "scalar_busy_per_wave": frac("SQ_ACTIVE_INST_SCA"),
            "waiting_per_wave": frac("SQ_WAIT_INST_ANY"), "waves": g("SQ_WAVES"),
            "instructions_per_wave": {k: (g("SQ_INSTS_" + k) / g("SQ_WAVES")) if g("SQ_INSTS_" + k) is not None and g("SQ_WAVES") else None
                                      for k in ("VALU", "SALU", "LDS", "VMEM_RD", "VMEM_WR", "SMEM")},
            "f64_fma_per_wave": (g("SQ_INSTS_VALU_FMA_F64") / g("SQ_WAVES")) if g("SQ_INSTS_VALU_FMA_F64") is not None and g("SQ_WAVES") else None,
            "mfma_f64_ops": g("SQ_INSTS_VALU_MFMA_MOPS_F64"), "lds_bank_conflict_cycles": g("SQ_LDS_BANK_CONFLICT"),
            "l2_hit_rate": (hit / (hit + miss)) if hit is not None and miss is not None and hit + miss > 0 else None,
            "note": "busy = cycles the unit executed this kernel's instructions / cycles its wavefronts were resident (per wavefront; "
                    "with w wavefronts per SIMD the SIMD's utilisation is about w x that)"}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _traffic(dom, tag):
    """HBM bytes of the dominant kernel per launch, from the committed rocprofv3 --pmc summary of this command
    (tools/pmc_profile.sh).  The summary is stamped with the digest of the kernel sources it was measured on; a stale one is
    refused rather than quoted."""
    import glob
    cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_%s.json" % tag)))
    if not cand:
        return None, None, "no PMC summary committed for this workload"
    pm = json.load(open(cand[-1]))
    if pm.get("kernel_source_hash") != kernel_source_hash():
        return None, os.path.basename(cand[-1]), "PMC summary is stale: kernel sources changed since it was measured"
    try:
        # KB -> bytes; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; "a + b": two kernels timed as one slot
        # (nlmpc_sqp_wg is launched twice per solve where the plan cut the working set's capacity: the solve, and a second pass in which every
        # workgroup whose instance did not overflow returns at once -- the solve's traffic is the larger launch's)
        stat = "max" if dom.startswith("nlmpc_sqp_wg") else "mean"
        return sum((2.0 * pm["FETCH_SIZE"][k][stat] + pm["WRITE_SIZE"][k][stat]) * 1024.0 for k in dom.split(" + ")), os.path.basename(cand[-1]), None
    except KeyError:
        return None, os.path.basename(cand[-1]), "kernel %s not in the PMC summary" % dom


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=None, choices=sorted(CONFIGS), help="BASELINE.json config number (1..5)")
    ap.add_argument("--workload", default=None, choices=["lmpc", "lmpc50", "lmpc-hetero", "ugv", "osc8", "osc6", "vanderpol"],
                    help="lmpc = quadrotor LMPC N=20 (config 2, the default); lmpc50 = N=50 (config 4); the others are NLMPC")
    ap.add_argument("--batch", type=int, default=None, help="instances per GPU")
    ap.add_argument("--horizon", type=int, default=None, help="LMPC prediction horizon (overrides the workload's)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline sample budget (0 = skip)")
    ap.add_argument("--nlmpc-extra", type=int, default=1, help="default (LMPC) workload: append the NLMPC configs' quick figures as the `nlmpc` key (0 = skip)")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the timed steps rotate over (each with its own controller handle, workspace and outputs); "
                         "1 = strictly serial steps (default: what `value`, the roofline and the rocprof trace refer to)")
    ap.add_argument("--pipeline-streams", type=int, default=3,
                    help="extra leg after the timed region: the same steps rotated over this many streams, reported as "
                         "`pipelined` (0 = skip)")
    args = ap.parse_args()
    workload, batch = CONFIGS[args.config] if args.config else ("lmpc", 4096)
    if args.workload:
        workload = args.workload
        batch = {"lmpc": 4096, "lmpc50": 32768, "lmpc-hetero": 4096, "ugv": 4096, "osc8": 1024, "osc6": 1024, "vanderpol": 4096}[workload]
    if args.batch:
        batch = args.batch
    nl = workload not in ("lmpc", "lmpc50", "lmpc-hetero")
    steps = args.steps if args.steps is not None else (200 if not nl else {"vanderpol": 50, "ugv": 5, "osc6": 5, "osc8": 3}[workload])
    warmup = args.warmup if args.warmup is not None else (20 if not nl else 1)

    # ---- ranks: the driver starts them (RANK / WORLD_SIZE in the environment) or we do
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.exit("bench.py --gpus %d: only %d GPU(s) visible" % (args.gpus, have))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    import torch.distributed as dist
    # MPCX_FORCE_DIST=1: take the RCCL path (communicator, all-gather, barriers) even with one rank -- a single-GPU box can
    # then exercise the code the multi-GPU launches run
    use_dist = world > 1 or os.environ.get("MPCX_FORCE_DIST") == "1"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    gather = None
    if use_dist:
        from libmpc_amd.distributed import ControlGather
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)      # rendezvous, barriers, the time reduction
            assert dist.get_world_size() == args.gpus
        gather = ControlGather(local, rank, world)           # the data-path collective: RCCL behind the C ABI
        assert gather.world == world

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    B = batch
    if nl:
        run_nlmpc(args, workload, B, steps, warmup, world, rank, local, dev, gather, barrier)
    elif workload == "lmpc-hetero":
        run_lmpc_hetero(args, args.horizon if args.horizon else 20, B, steps, warmup, world, rank, local, dev, gather, barrier)
    else:
        ph = args.horizon if args.horizon else (50 if workload == "lmpc50" else 20)
        run_lmpc(args, ph, B, steps, warmup, world, rank, local, dev, gather, barrier)
    if world > 1:
        dist.destroy_process_group()


def timed(step, steps, warmup, barrier, world, dev):
    import torch.distributed as dist
    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def multi_gpu_configs(world, rank, local, dev, gather, barrier):
    """BASELINE.json's two multi-GPU configs in the line of the DEFAULT multi-GPU command (`bench.py --gpus N`, which weak-scales the headline
    config 2): config 4 (quadrotor LMPC N = 50, 32768 instances per rank) and config 5 (eight oscillators, 1024 per rank), batch-sharded with the
    RCCL all-gather of u* behind every solve, timed as the headline is (barrier, K steps, barrier, maximum over the ranks).  Every rank runs this;
    rank 0 files the figures under `multi_gpu_configs` -- a scaling run of the default command then yields the curves north_star asks for."""
    import ctypes as C
    from libmpc_amd._capi import check
    from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc
    res = {}
    s = torch.cuda.current_stream(local)
    # config 4
    B4, steps4 = 32768, 10
    c4 = quadrotor_lmpc(50, device=local)
    x0, u0, yref = quadrotor_batch(B4, first=rank * B4)
    b4, r4, keep4 = c4.make_batch(x0, u0, yref=yref)
    all4 = torch.empty((world * B4, 4), dtype=torch.float64, device=dev)

    def step4():
        c4.launch(b4, s)
        gather.allgather(r4.cmd, out=all4, stream=s.cuda_stream)

    dt4 = timed(step4, steps4, 2, barrier, world, dev)
    assert torch.equal(all4[rank * B4:(rank + 1) * B4], r4.cmd)
    res["config4_lmpc50_b32768_per_gpu"] = {"value": world * B4 * steps4 / dt4, "unit": "solves/s", "ms_per_step": dt4 / steps4 * 1e3, "steps": steps4,
                                            "total_batch": world * B4, "solved_fraction": float((r4.status == 0).double().mean()),
                                            "workload": "quadrotor LMPC nx=12 nu=4 ph=ch=50, %d instances per rank, RCCL all-gather of u* (%d x 4 doubles)" % (B4, world * B4)}
    del c4, b4, r4, keep4, all4
    # config 5
    B5, steps5 = 1024, 3
    c5, x5, u5 = nl_make("osc8", B5, first=rank * 7919, device=local)
    b5, o5 = c5.make_batch(torch.from_numpy(x5), torch.from_numpy(u5))
    all5 = torch.empty((world * B5, c5.nu), dtype=torch.float64, device=dev)

    def step5():
        check(c5._lib.mpcx_nlmpc_solve_batch(c5._h, C.byref(b5), s.cuda_stream))
        gather.allgather(o5["cmd"], out=all5, stream=s.cuda_stream)

    dt5 = timed(step5, steps5, 1, barrier, world, dev)
    assert torch.equal(all5[rank * B5:(rank + 1) * B5], o5["cmd"])
    st5 = o5["solver_status"].cpu().numpy()
    res["config5_osc8_b1024_per_gpu"] = {"value": world * B5 * steps5 / dt5, "unit": "solves/s", "ms_per_step": dt5 / steps5 * 1e3, "steps": steps5,
                                         "total_batch": world * B5, "solved_fraction": float(np.isin(st5, (3, 4)).mean()),
                                         "workload": "networked oscillators NLMPC, 8 oscillators nx=16 nu=8 ph=30 ch=15, %d instances per rank, RCCL all-gather of u* (%d x 8 doubles)" % (B5, world * B5)}
    return res


def run_lmpc(args, ph, B, steps, warmup, world, rank, local, dev, gather, barrier):
    from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc

    x0, u0, yref = quadrotor_batch(B, first=rank * B)
    ns = max(1, args.streams)
    lanes = []
    for k in range(ns):
        c_k = quadrotor_lmpc(ph, device=local)
        b_k, r_k, keep_k = c_k.make_batch(x0, u0, yref=yref)
        all_k = torch.empty((world * B, 4), dtype=torch.float64, device=dev) if gather else None
        lanes.append((c_k, b_k, r_k, keep_k, torch.cuda.current_stream(local) if ns == 1 else torch.cuda.Stream(device=dev), all_k))
    ctl, batch, res, keep, stream, _ = lanes[0]
    info = ctl.info()
    counter = [0]

    def step():
        c_k, b_k, r_k, _, s_k, all_k = lanes[counter[0] % ns]
        counter[0] += 1
        with torch.cuda.stream(s_k):
            c_k.launch(b_k, s_k)
            if gather:
                return gather.allgather(r_k.cmd, out=all_k, stream=s_k.cuda_stream)     # same stream: starts when the solve retires
        return r_k.cmd

    dt = timed(step, steps, warmup, barrier, world, dev)
    serial = None
    if gather:
        # The collective off the solve stream: step k's all-gather (latency-bound, tens of microseconds over xGMI) travels on a second
        # stream behind an event while step k+1's solve already runs (libmpc_amd.distributed.OverlappedGather, two result buffers).  K
        # steps, every one with its solve and its all-gather inside a timed region of its own, reported beside `value`.
        from libmpc_amd.distributed import OverlappedGather
        b2, r2, keep2 = ctl.make_batch(x0, u0, yref=yref)
        og = OverlappedGather(B, 4, dev, gather=gather, solve_stream=stream)
        og.cmd = [res.cmd, r2.cmd]
        pair = [batch, b2]
        kcount = [0]

        def step_overlapped():
            og.step(kcount[0], lambda i, s: ctl.launch(pair[i], s))
            kcount[0] += 1

        dt_o = timed(step_overlapped, steps, warmup, barrier, world, dev)
        og.finish()
        last = kcount[0] - 1
        assert torch.equal(og.gathered(last)[rank * B:(rank + 1) * B], og.cmd[last % 2])      # this rank's rows of the last gather are its own results
        # `value` stays the in-series figure: on one rank (and wherever the collective is shorter than the five extra host calls per step
        # that ordering two streams takes) the overlap does not pay at a 50 us step; the overlapped figure is reported beside it
        serial = {"value": world * B * steps / dt_o, "ms_per_step": dt_o / steps * 1e3,
                  "note": "all-gather of step k on its own stream behind an event while step k+1 solves (two result buffers)"}
        # ... and the same overlap as ONE HIP graph per buffer parity: {solve k || all-gather k-1}, one launch per step instead of five host calls
        try:
            from libmpc_amd.distributed import GraphedOverlap
            go = GraphedOverlap(og, lambda i, s: ctl.launch(pair[i], s))
            dt_g = timed(go.step, steps, warmup, barrier, world, dev)
            last_all = go.flush()
            i_last = (go.k - 1) % 2
            assert torch.equal(last_all[rank * B:(rank + 1) * B], og.cmd[i_last])
            serial["graph"] = {"value": world * B * steps / dt_g, "ms_per_step": dt_g / steps * 1e3,
                               "note": "{solve k || all-gather k-1} captured as one HIP graph per buffer parity, one graph launch per step"}
        except Exception as e:      # (a runtime whose RCCL cannot be captured: the eager figures above stand)
            serial["graph"] = {"value": None, "note": "graph capture of the overlapped step failed: %s" % str(e)[:200]}

    # the default multi-GPU command also times BASELINE's two multi-GPU configs (every rank takes part: collectives inside)
    mgc = multi_gpu_configs(world, rank, local, dev, gather, barrier) if (gather and args.config is None and args.workload is None and not args.batch and not args.horizon) else None

    # extra leg: consecutive batches are independent, so a serving loop keeps several in flight -- the tail of one launch
    # (it lasts as long as its slowest instance) overlaps the start of the next.  Not `value`: reported beside it.
    pipelined = None
    if world == 1 and args.pipeline_streams > 1:
        pl = []
        for k in range(args.pipeline_streams):
            c_k = quadrotor_lmpc(ph, device=local)
            b_k, r_k, keep_k = c_k.make_batch(x0, u0, yref=yref)
            pl.append((c_k, b_k, r_k, keep_k, torch.cuda.Stream(device=dev)))
        for i in range(3 * len(pl)):
            pl[i % len(pl)][0].launch(pl[i % len(pl)][1], pl[i % len(pl)][4])
        torch.cuda.synchronize()
        tp0 = time.perf_counter()
        for i in range(steps):
            c_k, b_k, _, _, s_k = pl[i % len(pl)]
            c_k.launch(b_k, s_k)
        torch.cuda.synchronize()
        pipelined = {"value": B * steps / (time.perf_counter() - tp0), "unit": "solves/s", "streams": len(pl),
                     "note": "same steps, independent handles and buffers per stream, launches overlap"}

    # per-step latency distribution (host-synchronised single steps), outside the timed region
    lat = []
    for _ in range(min(50, max(5, steps))):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t1)
    lat_p50 = float(np.median(lat)) * 1e3
    # the same single steps as one HIP graph each (mpcx_lmpc_graph_*: queue reset + three kernels behind one launch)
    lat_graph = None
    if world == 1:
        side = torch.cuda.Stream(device=dev)
        g = ctl.make_graph(batch, side)
        lg = []
        for _ in range(min(50, max(5, steps))):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ctl.launch_graph(g, side)
            side.synchronize()
            lg.append(time.perf_counter() - t1)
        ctl.destroy_graph(g)
        lat_graph = float(np.median(lg)) * 1e3

    if rank != 0:
        return
    # per-kernel average launch duration, HIP events on the launch stream (each kernel timed alone)
    import ctypes as C
    reps = max(10, min(steps, 100))
    ms3 = (C.c_float * 3)()
    ctl._lib.mpcx_lmpc_debug_time_kernels(ctl._h, C.byref(batch), C.c_void_p(stream.cuda_stream), reps, ms3)
    all_ms = ctl.time_launches(batch, reps, stream)
    torch.cuda.synchronize()
    iters = res.iterations.cpu().numpy().astype(np.float64)
    rounds = res.polish_rounds.cpu().numpy().astype(np.float64)
    na = res.active_count.cpu().numpy().astype(np.float64)
    status = res.status.cpu().numpy()
    nz, mg = float(info["nz"]), float(info["mg"])
    nin = 12 + 4 + 12 + 1
    # algorithmic flops: what the arithmetic needs, not what padding executes
    fl_assemble = B * (2.0 * (nz + mg + nin + nin) * nin + 2.0 * (nz + mg) * nz)
    fl_polish = float((rounds * (na ** 3 / 3.0 + 2.0 * na ** 2 + 2.0 * na * (nz + mg) + 8.0 * (nz + mg))).sum() + B * 4.0 * nz)
    fl_admm = float(iters.sum() * info["flops_per_admm_iter"])
    if ms3[0] < 1e-3:      # one-kernel form (lmpc_solve_group: MFMA assemble into LDS + one wavefront per instance): both parts' flops
        kern = {"lmpc_solve_group": (ms3[1], fl_assemble + fl_polish), "lmpc_solve_admm": (ms3[2], fl_admm)}
    elif ctl.debug_get("flags")[0] != 0:
        # cost from its definition (ill-conditioned Hessians): lmpc_solve leaves w, lmpc_cost_mfma computes 0.5 w'Hw + f'w for
        # sixteen instances per pass over H -- the two are launched and timed back to back as one slot
        kern = {"lmpc_assemble_mfma": (ms3[0], fl_assemble), "lmpc_solve + lmpc_cost_mfma": (ms3[1], fl_polish + B * (2.0 * nz * nz + 2.0 * nz)),
                "lmpc_solve_admm": (ms3[2], fl_admm)}
    else:
        kern = {"lmpc_assemble_mfma": (ms3[0], fl_assemble), "lmpc_solve": (ms3[1], fl_polish), "lmpc_solve_admm": (ms3[2], fl_admm)}
    dom = max(kern, key=lambda k: kern[k][0])
    kern_ms, flops = kern[dom]
    bytes_alg = float(B) * (8.0 * (12 + 4 + 12) + 8.0 * 4 + 8.0 + 16.0)   # x0,u0,yref in; cmd,cost,4 ints out
    ach_tf = flops / (kern_ms * 1e-3) / 1e12
    traffic, traffic_src, traffic_note = _traffic(dom, "lmpc%d_b%d" % (ph, B))
    roof = {"bound": "mfma", "achieved": ach_tf, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
            "frac": ach_tf / PEAK_FP64_TFLOPS, "traffic": traffic, "traffic_source": traffic_src, "traffic_note": traffic_note,
            "traffic_over_algorithmic": (traffic / bytes_alg) if traffic else None,
            "kernel": dom, "kernel_ms": kern_ms, "algorithmic_flops_per_launch": flops,
            "note": "f64 path, latency/issue-bound small dense factorisations; peak = FP64 vector = FP64 MFMA peak",
            "kernels_ms": {k: round(v[0], 5) for k, v in kern.items()}, "all_kernels_ms": all_ms,
            "all_kernels_achieved_TFLOPs": (fl_assemble + fl_polish + fl_admm) / (all_ms * 1e-3) / 1e12,
            "mean_polish_rounds": float(rounds.mean()), "mean_active_set": float(na.mean()),
            "mean_admm_iters": float(iters.mean()),
            "hbm_achieved_GBs": bytes_alg / (all_ms * 1e-3) / 1e9,
            "hbm_frac": bytes_alg / (all_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
            "algorithmic_bytes_per_launch": bytes_alg, "kernel_source_hash": kernel_source_hash(),
            "counters": _sq_counters(dom.split(" + ")[0], "lmpc%d_b%d" % (ph, B))}
    cpu = None
    if world == 1 and args.cpu_seconds > 0:
        # (i) one thread, every instance in turn -> per-solve latency and single-core rate; (ii) all host
        # cores, instances split over worker processes -> the node's CPU rate (`value`, `cores`)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import quadrotor_oracle
        o = quadrotor_oracle(ph)
        probe = o.solve_batch_constref(x0[:32], u0[:32], yref[:32])
        per = probe["seconds"] / 32
        n1 = int(max(32, min(B, 0.4 * args.cpu_seconds / per)))
        rr = o.solve_batch_constref(x0[:n1], u0[:n1], yref[:n1])
        ps = np.sort(rr["per_solve_seconds"])
        ncores = _usable_cores()
        import multiprocessing as mp
        per_core = max(8, min(64, B // ncores))
        budget = 0.5 * args.cpu_seconds
        with mp.get_context("fork").Pool(ncores) as pool:
            chunks = [(ph, x0[(i * per_core) % B:][:per_core], u0[(i * per_core) % B:][:per_core], yref[(i * per_core) % B:][:per_core], budget)
                      for i in range(ncores)]
            res_cpu = pool.map(_cpu_worker, chunks)
        done = sum(r[0] for r in res_cpu)
        t_all = max(r[1] for r in res_cpu)         # workers run concurrently: the slowest one closes the interval
        cpu = {"value": done / t_all, "unit": "solves/s", "cores": ncores, "kind": "port",
               "sample": f"{done} solves in {t_all:.1f} s: {ncores} worker processes (one per host core), each repeating its own "
                         f"{per_core} instances of the same batch, set-up per solve as LOptimizer::run; single-thread "
                         f"figures from the first {n1} instances",
               "single_thread_value": n1 / rr["seconds"],
               "p50_ms": float(ps[len(ps) // 2] * 1e3), "p99_ms": float(ps[int(len(ps) * 0.99) - 1] * 1e3)}
    total = world * B * steps
    out = {"metric": "MPC solves/sec (whole node) + p50 solve latency, quadrotor LMPC N=%d batch=%d" % (ph, B),
           "value": total / dt, "unit": "solves/s", "n_gpus": world, "steps": steps, "warmup": warmup,
           "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "quadrotor_ex.cpp LMPC nx=12 nu=4 ny=12 ph=ch=%d, batch %d per GPU, "
                                  "SplitMix64 x0/u0/yref (SURVEY 8d), maximum_iteration=250" % (ph, B),
                      "parallelism": ("batch-sharded x%d, RCCL all-gather of u* (%d x 4 doubles) through mpcx_allgather_u"
                                      % (world, world * B)) if gather else "single GPU",
                      "rccl_ranks": gather.world if gather else 0, "streams": ns},
           "p50_step_latency_ms": lat_p50,
           "p50_step_latency_graph_ms": lat_graph,
           "pipelined": pipelined,
           "allgather_overlapped": serial,
           "solved_fraction": float((status == 0).mean()),
           "roofline": roof, "cpu_baseline": cpu}
    if mgc:
        out["multi_gpu_configs"] = mgc
    if world == 1 and args.nlmpc_extra:
        out["nlmpc"] = nlmpc_extra(local)
    print(json.dumps(out))


def run_lmpc_hetero(args, ph, B, steps, warmup, world, rank, local, dev, gather, barrier):
    """every instance its own controller (own dynamics, weights, limits: quadrotor_variant(k)) -- SURVEY.md 8(d)'s "fully
    heterogeneous" case.  One step = one batched solve of B instances against B models; the bank's set-up (device condensing) is
    timed apart: in the reference it is part of every solve, here it is paid when a model changes."""
    import ctypes as C
    from libmpc_amd import LMPCHetero
    from libmpc_amd.workloads import quadrotor_batch, quadrotor_variant
    t0 = time.perf_counter()
    ctrls = [quadrotor_variant(rank * B + k, ph, device=-1) for k in range(B)]
    t_cfg = time.perf_counter() - t0
    het = LMPCHetero(ctrls, device=local)
    flags = het.debug_get(0, "flags")
    x0, u0, yref = quadrotor_batch(B, first=rank * B)
    b, res, keep, mi = het.make_batch(x0, u0, yref=yref)
    stream = torch.cuda.current_stream(local)
    all_k = torch.empty((world * B, 4), dtype=torch.float64, device=dev) if gather else None

    def step():
        het.launch(b, mi, stream)
        if gather:
            return gather.allgather(res.cmd, out=all_k, stream=stream.cuda_stream)
        return res.cmd

    dt = timed(step, steps, warmup, barrier, world, dev)
    if rank != 0:
        return
    ms = het.time_launches(b, mi, max(10, min(steps, 100)), stream)
    torch.cuda.synchronize()
    status = res.status.cpu().numpy()
    rounds = res.polish_rounds.cpu().numpy().astype(np.float64)
    nx, nu, ny = 12, 4, 12
    # compulsory bytes of a fully heterogeneous solve -- model, per-step weights, bounds and references in, u* out
    bytes_in = 8.0 * (nx * nx + nx * nu + ny * nx) + 8.0 * (ny + 2 * nu) * ph + 8.0 * 2 * (nx + ny + nu) * ph + 8.0 * (ny + 2 * nu) * ph + 8.0 * (nx + nu)
    bytes_alg = float(B) * (bytes_in + 8.0 * nu + 8.0)
    gbs = bytes_alg / (ms * 1e-3) / 1e9
    # the factors a solve fetches from its own controller: mpcx_lmpc_hetero_get_info counts the first nz columns of Y = [Hinv; G Hinv];
    # the assemble kernel reads the Hinv rows only (G t0 comes from a roll-out of the model it has in LDS)
    ti = het._template.info()
    fbytes = het.bytes_per_model - 8.0 * ti["nz"] * ((ti["mg"] + 1) // 2 * 2)
    gbs_read = float(B) * fbytes / (ms * 1e-3) / 1e9
    traffic, traffic_src, traffic_note = _traffic("lmpc_assemble_generic", "lmpchetero%d_b%d" % (ph, B))
    roof = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
            "traffic": traffic, "traffic_source": traffic_src, "traffic_note": traffic_note,
            "traffic_over_algorithmic": (traffic / bytes_alg) if traffic else None,
            "kernel": "lmpc_assemble_generic + lmpc_solve (every instance reads its own model's factors)", "all_kernels_ms": ms,
            "algorithmic_bytes_per_solve": bytes_in + 8.0 * nu + 8.0, "algorithmic_bytes_per_launch": bytes_alg,
            "factor_bytes_read_per_solve": fbytes, "factor_read_GBs": gbs_read, "factor_read_frac": gbs_read / PEAK_HBM_GBS,
            "mean_polish_rounds": float(rounds.mean()),
            "note": "algorithmic = what the reference's ProblemBuilder consumes per solve (SURVEY 8d: 18 176 B at N = 20); this path reads the "
                    "controller's precomputed factors instead (-Hinv and the model matrices: factor_bytes_read_per_solve) and pays the condensing once per "
                    "model change (set_up below)",
            "kernel_source_hash": kernel_source_hash(), "counters": _sq_counters("lmpc_assemble_generic", "lmpchetero%d_b%d" % (ph, B))}
    setup = {"controllers": B, "condensed_on_device": bool(flags[1]), "condense_kernel_ms": float(flags[2]), "create_ms": float(flags[3]),
             "python_configure_s": t_cfg, "flops_per_controller": float(flags[4]),
             "mfma_achieved_TFLOPs": (B * flags[4] / (flags[2] * 1e-3) / 1e12) if flags[2] > 0 else None,
             "mfma_frac": (B * flags[4] / (flags[2] * 1e-3) / 1e12 / PEAK_FP64_TFLOPS) if flags[2] > 0 else None,
             "note": "lmpc_condense_models: one workgroup per controller, prediction matrices / Hessian / dual Hessian / ADMM matrix on "
                     "v_mfma_f64_16x16x4_f64, Cholesky and triangular inverses in LDS; flops = the host set-up's count"}
    cpu = None
    if world == 1 and args.cpu_seconds > 0:
        import multiprocessing as mp
        ncores = _usable_cores()
        per_core = max(8, min(64, B // ncores))
        x0n, u0n, yrn = (np.asarray(a.cpu().numpy() if hasattr(a, "cpu") else a) for a in (x0, u0, yref))
        jobs = [(ph, (i * per_core) % B, x0n[(i * per_core) % B:][:per_core], u0n[(i * per_core) % B:][:per_core], yrn[(i * per_core) % B:][:per_core],
                 0.5 * args.cpu_seconds) for i in range(ncores)]
        with mp.get_context("fork").Pool(ncores) as pool:
            res_cpu = pool.map(_hetero_cpu_worker, jobs)
        done = sum(r[0] for r in res_cpu); t_all = max(r[1] for r in res_cpu)
        cpu = {"value": done / t_all, "unit": "solves/s", "cores": ncores, "kind": "port",
               "sample": f"{done} solves in {t_all:.1f} s: {ncores} worker processes (one per host core), each on its own {per_core} instances of the same "
                         f"batch, every instance on the C oracle of its own controller (one controller object per problem, configured outside the "
                         f"timed loop; set-up per solve as LOptimizer::run)"}
    total = world * B * steps
    out = {"metric": "MPC solves/sec (whole node), quadrotor LMPC N=%d, every instance its own controller, batch=%d" % (ph, B),
           "value": total / dt, "unit": "solves/s", "n_gpus": world, "steps": steps, "warmup": warmup,
           "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "quadrotor LMPC nx=12 nu=4 ny=12 ph=ch=%d, %d controllers (quadrotor_variant: own dynamics, weights, limits), one "
                                  "instance each per GPU, SplitMix64 x0/u0/yref" % (ph, B),
                      "parallelism": ("batch-sharded x%d" % world) if gather else "single GPU", "rccl_ranks": gather.world if gather else 0},
           "solved_fraction": float((status == 0).mean()), "roofline": roof, "set_up": setup, "cpu_baseline": cpu}
    print(json.dumps(out))


# ---------------------------------------------------------------------------------------------------------------------
def nl_make(name, B, first=0, device=0):
    """the synthetic batches of SURVEY.md 8(d) configs 1, 3, 5 (same generator as tools/nlmpc_bench.py); `first` offsets
    the stream so that every rank solves different instances"""
    from libmpc_amd.nlmpc import NLMPC, NLParameters, OSCILLATORS6, OSCILLATORS8, UGV, VANDERPOL
    rng = np.random.default_rng(first)
    if name == "ugv":
        c = NLMPC(UGV, 30, 30, 0.1, device=device)
        c.setOptimizerParameters(NLParameters(maximum_iteration=150, hard_constraints=0))
        x0 = np.zeros((B, 4)); x0[:, :2] = rng.uniform(-0.5, 0.5, size=(B, 2))
    elif name == "vanderpol":
        c = NLMPC(VANDERPOL, 10, 5, 0.1, device=device)
        c.setOptimizerParameters(NLParameters(maximum_iteration=200))
        x0 = rng.uniform(-1, 1, size=(B, 2))
    else:
        n = int(name[3:])
        c = NLMPC(OSCILLATORS6 if n == 6 else OSCILLATORS8, 20 if n == 6 else 30, 10 if n == 6 else 15, 0.1, device=device)
        c.setOptimizerParameters(NLParameters(maximum_iteration=200))
        x0 = rng.uniform(-0.1, 0.1, size=(B, 2 * n)); x0[:, 0] += 1.0
    return c, x0, np.zeros((B, c.nu))


def nl_flops(c, name, it, nact, wg=True):
    """algorithmic flops of the SQP solves of one batch, for the path the kernel takes: per iteration
    the transcription's function evaluations, the condensing, the reduction, the BFGS update, the dual active-set steps
    the final active set needs, the state step and one line-search round.  Where no sub-problem row reads a state
    (Van der Pol, the oscillator networks) the sensitivities Phi are never formed: condensing, reduction and the state
    step are single sweeps over the dynamics blocks, and the sub-problem's rows are one-entry lists -- counted as such."""
    nx, nu, ph, ch, nz = c.nx, c.nu, c.ph, c.ch, c.nz
    nzu = ch * nu; nq = nzu + 1; nxs = ph * nx
    ct = name != "ugv"
    matrix_free = name != "ugv"
    f_f = {"vanderpol": 8.0, "ugv": 16.0}.get(name, 6.0 * nu + 2.0 * nu * nu)        # one vector-field call
    f_cost = (ph + 1) * (2.0 * nx + 2.0 * nu)                                          # one cost call
    f_ineq = 4.0 if name == "ugv" else 1.0                                             # one constraint component
    ev = f_cost * (ph * (nx + nu) + 3) + f_f * ph * (1 + ct) * (1 + 2 * (nx + nu)) + f_ineq * c.nineq * 3
    sweep = ph * (2.0 * nx * nx * (1 + ct) + 2.0 * nx * nu)                            # one column through the dynamics blocks
    einv = ph * 2.0 * nx ** 3 if ct else 0.0
    if matrix_free:
        cond = sweep + einv
        red = sweep                                                                    # backward sweep + Ju' lam
        step = sweep
        qp = 2.0 * nq * nq + nact * (2.0 * nq * nact + 2.0 * nact * nact)              # B^-1 g; per step: z = v - V rr, two substitutions
    else:
        # rows of the sub-problem read states (config 3): the kernel condenses row by row, backwards (nlmpc_sqp_wg.hpp, condense_phi): one
        # adjoint sweep per dense row over the steps below the state row it reads (half the horizon on average: l <- Abar' l, Bbar' l and
        # cbar' l per step), one full chain for the reduced gradient and its inputs' part, the state step one more chain with the inputs
        # applied -- counted as that, not as the products with a stored Phi (61 column sweeps + two nz x nzu products) of rounds 1 to 4
        row = 0.5 * ph * (2.0 * nx * nx + 2.0 * nx * nu + 2.0 * nx)
        cond = c.nineq * row + sweep + einv
        red = 2.0 * nx * nzu
        step = sweep + 2.0 * nxs * nu
        qp = 2.0 * nq * nq + nact * (2.0 * nq * nq + 4.0 * nq * nact)
    bfgs = 8.0 * nq * nq
    ls = 8 * (f_cost + f_f * ph * (1 + ct) + f_ineq * c.nineq)
    # once per solve (the workgroup form: WgSqp::init_curvature): the sensitivities one step on and Qx Phi (2 nx^2 nzu each per step), Phi' (Qx Phi) by its
    # lower triangle (nx nzu^2 per step), the sweep that inverts the nq x nq matrix (nq^3)
    curv = ph * (4.0 * nx * nx * nzu + 1.0 * nx * nzu * nzu) + float(nq) ** 3 if wg else 0.0
    return float((it * (ev + cond + red + bfgs + step + ls) + it * qp + curv).sum())


def nlmpc_extra(local):
    """Configs 1, 3 and 5 in the line the driver runs (outside `value` and the timed region, ~1 s of GPU time): solves/s of a few steps at the
    quoted batches, and the time of one batched solve at a batch the workgroup form holds resident (the latency a controller sees)."""
    import ctypes as C
    from libmpc_amd._capi import check
    res = {}
    stream = torch.cuda.current_stream(local)
    for key, name, B, steps in (("config1_vanderpol_b4096", "vanderpol", 4096, 10), ("config3_ugv_b4096", "ugv", 4096, 2), ("config5_osc8_b1024", "osc8", 1024, 1),
                                ("config3_ugv_b256_latency", "ugv", 256, 2), ("config5_osc8_b256_latency", "osc8", 256, 1)):
        c, x0, u0 = nl_make(name, B, device=local)
        b, out = c.make_batch(torch.from_numpy(x0), torch.from_numpy(u0))
        check(c._lib.mpcx_nlmpc_solve_batch(c._h, C.byref(b), stream.cuda_stream)); torch.cuda.synchronize()      # warm-up
        ms = c.time_launches(b, steps, stream.cuda_stream)
        torch.cuda.synchronize()
        st = out["solver_status"].cpu().numpy(); it = out["iterations"].cpu().numpy()
        form = int(c._lib.mpcx_nlmpc_last_form(c._h))
        res[key] = {"solves_per_s": B / (ms * 1e-3), "kernel_ms": ms, "kernel": "nlmpc_sqp_wg" if form > 0 else "nlmpc_sqp",
                    "wavefronts_per_instance": form if form > 0 else 1, "mean_iterations": float(it.mean()),
                    "solved_fraction": float(np.isin(st, (3, 4)).mean())}
    # the receding-horizon loop a fleet controller runs (examples/ugv_ex.cpp's closed loop for 256 vehicles): every tick one batched
    # solve from the shifted previous solution with the curvature estimate carried over, then one plant step; the first (cold) tick is not timed
    c, x0, u0 = nl_make("ugv", 256, device=local)
    x = torch.from_numpy(x0).to("cuda:%d" % local); u = torch.from_numpy(u0).to("cuda:%d" % local)
    Ts, z, its, ticks = 0.1, None, 0.0, 12
    for k in range(ticks + 1):
        if k == 1:
            torch.cuda.synchronize(); t0 = time.perf_counter(); its = 0.0
        r = c.optimizeBatch(x, u, z_warm=z, warm_curvature=z is not None)
        u = r["cmd"]
        x = torch.stack([x[:, 0] + Ts * x[:, 2] + 0.5 * Ts * Ts * u[:, 0], x[:, 1] + Ts * x[:, 3] + 0.5 * Ts * Ts * u[:, 1],
                         x[:, 2] + Ts * u[:, 0], x[:, 3] + Ts * u[:, 1]], dim=1)
        z = r["z"]
        its += float(r["iterations"].float().mean())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res["config3_ugv_b256_closed_loop"] = {"ms_per_tick": dt / ticks * 1e3, "solves_per_s": 256 * ticks / dt, "mean_iterations": its / ticks,
                                           "not_failed": float((r["status"] != 3).float().mean()), "ticks": ticks,
                                           "note": "256 instances of config 3's synthetic batch in closed loop; warm start = shifted previous solution + carried curvature; "
                                                   "host-synchronised wall time per tick, plant step included"}
    return res


def run_nlmpc(args, name, B, steps, warmup, world, rank, local, dev, gather, barrier):
    c, x0, u0 = nl_make(name, B, first=rank * 7919, device=local)
    x0t, u0t = torch.from_numpy(x0), torch.from_numpy(u0)
    b, out = c.make_batch(x0t, u0t, multipliers=True)
    stream = torch.cuda.current_stream(local)
    all_u = torch.empty((world * B, c.nu), dtype=torch.float64, device=dev) if gather else None
    import ctypes as C
    from libmpc_amd._capi import check

    def step():
        check(c._lib.mpcx_nlmpc_solve_batch(c._h, C.byref(b), stream.cuda_stream))
        if gather:
            gather.allgather(out["cmd"], out=all_u, stream=stream.cuda_stream)

    dt = timed(step, steps, warmup, barrier, world, dev)
    if rank != 0:
        return
    kern_ms = c.time_launches(b, max(1, min(steps, 5)), stream.cuda_stream)       # HIP events on the launch stream
    torch.cuda.synchronize()
    st = out["solver_status"].cpu().numpy(); it = out["iterations"].cpu().numpy().astype(np.float64)
    nact = (out["multipliers"].cpu().numpy() != 0).sum(axis=1).astype(np.float64)
    form = int(c._lib.mpcx_nlmpc_last_form(c._h))
    flops = nl_flops(c, name, it, nact, wg=form > 0)
    bytes_alg = float(B) * 8.0 * (c.nx + c.nu + c.nu + 1 + 2)
    ach = flops / (kern_ms * 1e-3) / 1e12
    form = int(c._lib.mpcx_nlmpc_last_form(c._h))
    dom = "nlmpc_sqp_wg" if form > 0 else "nlmpc_sqp"
    traffic, traffic_src, traffic_note = _traffic(dom, "%s_b%d" % (name, B))
    roof = {"bound": "mfma", "achieved": ach, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP64_TFLOPS,
            "traffic": traffic, "traffic_source": traffic_src, "traffic_note": traffic_note,
            "traffic_over_algorithmic": (traffic / bytes_alg) if traffic else None,
            "kernel": dom, "kernel_ms": kern_ms, "algorithmic_flops_per_launch": flops,
            "note": ("f64 SQP, one workgroup of %d wavefront(s) per instance, the reduced problem in LDS" % form if form > 0 else
                     "f64 SQP, one instance per wavefront, the reduced problem in an HBM workspace") + "; bound by the issue latency of dependent instructions; flop model in DESIGN.md section 6",
            "form": {"kernel": dom, "wavefronts_per_instance": form if form > 0 else 1},
            "mean_iterations": float(it.mean()), "max_iterations": int(it.max()), "mean_active_rows": float(nact.mean()),
            "hbm_achieved_GBs": bytes_alg / (kern_ms * 1e-3) / 1e9, "hbm_frac": bytes_alg / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
            "algorithmic_bytes_per_launch": bytes_alg, "workspace_bytes_per_instance": int(c.debug_workspace_bytes()),
            "kernel_source_hash": kernel_source_hash(), "counters": _sq_counters(dom, "%s_b%d" % (name, B))}
    cpu = None
    if world == 1 and args.cpu_seconds > 0:
        ncores = _usable_cores()
        per = {"vanderpol": 64, "ugv": 8, "osc6": 4, "osc8": 2}[name]
        import multiprocessing as mp
        budget = 0.8 * args.cpu_seconds
        with mp.get_context("fork").Pool(ncores) as pool:
            res_cpu = pool.map(_nl_cpu_worker, [(name, x0[(i * per) % B:][:per], u0[(i * per) % B:][:per], budget) for i in range(ncores)])
        done = sum(r[0] for r in res_cpu); t_all = max(r[1] for r in res_cpu)
        cpu = {"value": done / t_all, "unit": "solves/s", "cores": ncores, "kind": "port",
               "sample": f"{done} solves in {t_all:.1f} s: {ncores} worker processes (one per host core), each cycling over its own {per} "
                         f"instance(s) of the same batch; compiled baseline = the reference's callbacks restated in C "
                         f"(oracle/nlmpc_callbacks.c: forward / central differences as Objective.hpp:198-265, Constraints.hpp:641-905) "
                         f"driving SLSQP (scipy's compiled Kraft code, which NLopt's LD_SLSQP translates), cold starts, 150 iterations at most"}
    label = {"vanderpol": "vanderpol_ex.cpp NLMPC nx=2 nu=1 ph=10 ch=5 (config 1)",
             "ugv": "ugv_ex.cpp NLMPC nx=4 nu=2 ph=ch=30, soft constraints (config 3)",
             "osc6": "networked_oscillators_ex.cpp NLMPC 6 oscillators nx=12 nu=6 ph=20 ch=10 (the reference example)",
             "osc8": "networked_oscillators_ex.cpp NLMPC 8 oscillators nx=16 nu=8 ph=30 ch=15 (config 5)"}[name]
    o = {"metric": "MPC solves/sec (whole node), %s batch=%d" % (name, B), "value": world * B * steps / dt, "unit": "solves/s",
         "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
         "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
         "config": {"workload": label + ", batch %d per GPU, cold starts" % B,
                    "parallelism": ("batch-sharded x%d, RCCL all-gather of u* through mpcx_allgather_u" % world) if gather else "single GPU",
                    "rccl_ranks": gather.world if gather else 0},
         "solved_fraction": float(np.isin(st, (3, 4)).mean()),
         "solver_status_counts": {int(k): int((st == k).sum()) for k in np.unique(st)},
         "roofline": roof, "cpu_baseline": cpu}
    print(json.dumps(o))


if __name__ == "__main__":
    main()
