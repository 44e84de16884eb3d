"""Single-instance and small-batch latency of one solve call (launch to completion, host-synchronised), p50 over repeats."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc  # noqa: E402
from tools.nlmpc_bench import make  # noqa: E402


def p50(fn, n=200):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return float(np.median(ts) * 1e6)


for ph in (10, 20, 50):
    c = quadrotor_lmpc(ph, device=0)
    for B in (1, 64, 4096):
        x0, u0, yref = quadrotor_batch(B)
        b, r, k = c.make_batch(x0, u0, yref=yref)
        c.launch(b); torch.cuda.synchronize()
        print("LMPC quadrotor N=%d batch %d: p50 %.1f us (kernels only %.1f us)" % (ph, B, p50(lambda: c.launch(b)), c.time_launches(b, 50) * 1e3))
for name in ("vanderpol", "ugv"):
    for B in (1, 64):
        c, x0, u0 = make(name, B)
        b, out = c.make_batch(x0, u0)
        c.time_launches(b, 1)
        print("NLMPC %s batch %d: kernel %.1f us" % (name, B, c.time_launches(b, 5) * 1e3))
