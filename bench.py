"""bench.py -- MPC solves/sec on the reference's quadrotor LMPC (N=20), batch 4096 per GPU.

One "step" = one pass of the hot path (batched LOptimizer::run) over one batch of synthetic
instances already resident in HBM.  N>1: one process per GPU (torch.distributed / RCCL),
each rank solves its own shard, then one all-gather of u* over xGMI; weak scaling.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_FP64_TFLOPS = 78.6      # MI355X FP64 vector == FP64 matrix peak; see DESIGN.md
PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="instances per GPU")
    ap.add_argument("--horizon", type=int, default=20)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline sample budget (0 = skip)")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the timed steps rotate over (each with its own controller handle, workspace and outputs); "
                         "1 = strictly serial steps (default: what `value`, the roofline and the rocprof trace refer to)")
    ap.add_argument("--pipeline-streams", type=int, default=3,
                    help="extra leg after the timed region: the same steps rotated over this many streams, reported as "
                         "`pipelined_value` (0 = skip)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    # MPCX_FORCE_DIST=1: take the RCCL path (process group, all-gather, barriers) even with one rank -- a single-GPU box can
    # then exercise the code the multi-GPU launches run
    use_dist = world > 1 or os.environ.get("MPCX_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from libmpc_amd.distributed import allgather_controls
    from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc

    B, ph = args.batch, args.horizon
    x0, u0, yref = quadrotor_batch(B, first=rank * B)
    ns = max(1, args.streams)
    lanes = []
    for k in range(ns):
        c_k = quadrotor_lmpc(ph, device=local)
        b_k, r_k, keep_k = c_k.make_batch(x0, u0, yref=yref)
        lanes.append((c_k, b_k, r_k, keep_k, torch.cuda.current_stream(local) if ns == 1 else torch.cuda.Stream(device=dev)))
    ctl, batch, res, keep, stream = lanes[0]
    info = ctl.info()
    counter = [0]

    def step():
        c_k, b_k, r_k, _, s_k = lanes[counter[0] % ns]
        counter[0] += 1
        with torch.cuda.stream(s_k):
            c_k.launch(b_k, s_k)
            if use_dist:
                return allgather_controls(r_k.cmd, force=True)
        return r_k.cmd

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

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
        for i in range(args.steps):
            c_k, b_k, _, _, s_k = pl[i % len(pl)]
            c_k.launch(b_k, s_k)
        torch.cuda.synchronize()
        pipelined = {"value": B * args.steps / (time.perf_counter() - tp0), "unit": "solves/s", "streams": len(pl),
                     "note": "same steps, independent handles and buffers per stream, launches overlap"}

    # per-step latency distribution (host-synchronised single steps), outside the timed region
    lat = []
    for _ in range(min(50, max(5, args.steps))):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t1)
    lat_p50 = float(np.median(lat)) * 1e3

    if rank == 0:
        # per-kernel average launch duration, HIP events on the launch stream (each kernel timed alone)
        import ctypes as C
        reps = max(10, min(args.steps, 100))
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
        fl_polish = float((rounds * (na ** 3 / 3.0 + 2.0 * na ** 2 + 2.0 * na * (nz + mg) + 8.0 * (nz + mg))).sum()
                          + B * (2.0 * nz * nz + 2.0 * nz))
        fl_admm = float(iters.sum() * info["flops_per_admm_iter"])
        kern = {"lmpc_assemble_mfma": (ms3[0], fl_assemble), "lmpc_solve": (ms3[1], fl_polish), "lmpc_solve_admm": (ms3[2], fl_admm)}
        dom = max(kern, key=lambda k: kern[k][0])
        kern_ms, flops = kern[dom]
        bytes_alg = float(B) * (8.0 * (12 + 4 + 12) + 8.0 * 4 + 8.0 + 16.0)   # x0,u0,yref in; cmd,cost,4 ints out
        ach_tf = flops / (kern_ms * 1e-3) / 1e12
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; they come from
        # the two rocprofv3 --pmc passes of this same command whose per-kernel means are committed under profiles/
        # (tools/pmc_summary.py).  KB -> bytes; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950.
        traffic, traffic_src = None, None
        try:
            import glob
            cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
            if cand and B == 4096 and ph == 20:
                pm = json.load(open(cand[-1]))
                traffic = (2.0 * pm["FETCH_SIZE"][dom]["mean"] + pm["WRITE_SIZE"][dom]["mean"]) * 1024.0
                traffic_src = os.path.basename(cand[-1])
        except Exception:
            traffic = None
        roof = {"bound": "mfma", "achieved": ach_tf, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                "frac": ach_tf / PEAK_FP64_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                "kernel": dom, "kernel_ms": kern_ms, "algorithmic_flops_per_launch": flops,
                "note": "f64 path, latency/issue-bound small dense factorisations; peak = FP64 vector = FP64 MFMA peak",
                "kernels_ms": {k: round(v[0], 5) for k, v in kern.items()}, "all_kernels_ms": all_ms,
                "all_kernels_achieved_TFLOPs": (fl_assemble + fl_polish + fl_admm) / (all_ms * 1e-3) / 1e12,
                "mean_polish_rounds": float(rounds.mean()), "mean_active_set": float(na.mean()),
                "mean_admm_iters": float(iters.mean()),
                "hbm_achieved_GBs": bytes_alg / (all_ms * 1e-3) / 1e9,
                "hbm_frac": bytes_alg / (all_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                "algorithmic_bytes_per_launch": bytes_alg}
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
        total = world * B * args.steps
        out = {"metric": "MPC solves/sec (whole node) + p50 solve latency, quadrotor LMPC N=%d batch=%d" % (ph, B),
               "value": total / dt, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "quadrotor_ex.cpp LMPC nx=12 nu=4 ny=12 ph=ch=%d, batch %d per GPU, "
                                      "SplitMix64 x0/u0/yref (SURVEY 8d), maximum_iteration=250" % (ph, B),
                          "parallelism": "batch-sharded x%d, all-gather of u*" % world if world > 1 else "single GPU",
                          "streams": ns},
               "p50_step_latency_ms": lat_p50,
               "pipelined": pipelined,
               "solved_fraction": float((status == 0).mean()),
               "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
