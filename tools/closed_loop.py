"""Receding-horizon loop on the device: B quadrotor controllers, each tick = one batched solve +
one plant step x+ = A x + B u (the closed loop of examples/quadrotor_ex.cpp run for a batch), with and without the
warm start that carries the working set from tick to tick.  Usage: python tools/closed_loop.py [batch] [ticks]"""
import json
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc, quadrotor_matrices  # noqa: E402


def run(B, ticks, warm):
    c = quadrotor_lmpc(20, device=0)
    x0, u0, yref = quadrotor_batch(B)
    Ad, Bd, _ = quadrotor_matrices()
    A = torch.as_tensor(Ad).cuda().T.contiguous(); Bm = torch.as_tensor(Bd).cuda().T.contiguous()
    x = torch.as_tensor(x0).cuda(); u = torch.as_tensor(u0).cuda(); yr = torch.as_tensor(yref).cuda()
    prev = None
    rounds = 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(ticks):
        r = c.optimizeBatch(x, u, yref=yr, want_active=warm, warm=prev if warm else None, warm_shift=True)
        x = x @ A + r.cmd @ Bm
        u = r.cmd
        prev = r
        rounds += r.polish_rounds.float().mean()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(warm=warm, batch=B, ticks=ticks, solves_per_s=B * ticks / dt, ms_per_tick=dt / ticks * 1e3,
                mean_rounds=float(rounds) / ticks, solved=float((r.status == 0).float().mean()))


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    run(B, 5, True)
    for w in (False, True):
        print(json.dumps(run(B, ticks, w)))
