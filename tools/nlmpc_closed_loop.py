"""Receding-horizon NLMPC loop on the device: B UGV controllers, each tick = one batched solve + one
plant step (the closed loop of examples/ugv_ex.cpp for a batch), cold starts against the shifted warm start of
NLOptimizer::run (NLOptimizer.hpp:460-510).  Usage: python tools/nlmpc_closed_loop.py [batch] [ticks]"""
import json
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tools.nlmpc_bench import make  # noqa: E402


def run(B, ticks, warm, curvature=False):
    c, x0, u0 = make("ugv", B)
    Ts = 0.1
    x = x0.cuda(); u = u0.cuda()
    z = None
    its = 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(ticks):
        r = c.optimizeBatch(x, u, z_warm=z if warm else None, warm_curvature=curvature)
        u = r["cmd"]
        x = torch.stack([x[:, 0] + Ts * x[:, 2] + 0.5 * Ts * Ts * u[:, 0], x[:, 1] + Ts * x[:, 3] + 0.5 * Ts * Ts * u[:, 1],
                         x[:, 2] + Ts * u[:, 0], x[:, 3] + Ts * u[:, 1]], dim=1)
        z = r["z"]
        its += r["iterations"].float().mean()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = float((r["status"] != 3).float().mean())
    return dict(warm=warm, keep_curvature=curvature, batch=B, ticks=ticks, solves_per_s=B * ticks / dt, ms_per_tick=dt / ticks * 1e3,
                mean_iterations=float(its) / ticks, not_failed=ok)


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    for w, k in ((False, False), (True, False), (True, True)):
        print(json.dumps(run(B, ticks, w, k)))
