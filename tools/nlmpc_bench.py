"""Times the batched NLMPC solve (mpcx_nlmpc_solve_batch) on the synthetic batches of SURVEY.md 8(d) configs 3 and 5.
Usage: python tools/nlmpc_bench.py [ugv|osc6|osc8|vanderpol] [batch] [repeats]"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from libmpc_amd.nlmpc import NLMPC, NLParameters, OSCILLATORS6, OSCILLATORS8, UGV, VANDERPOL  # noqa: E402


def make(name, B, seed=0):
    rng = np.random.default_rng(seed)
    if name == "ugv":
        c = NLMPC(UGV, 30, 30, 0.1)
        c.setOptimizerParameters(NLParameters(maximum_iteration=150, hard_constraints=0))
        x0 = np.zeros((B, 4)); x0[:, :2] = rng.uniform(-0.5, 0.5, size=(B, 2))
    elif name == "vanderpol":
        c = NLMPC(VANDERPOL, 10, 5, 0.1)
        c.setOptimizerParameters(NLParameters(maximum_iteration=200))
        x0 = rng.uniform(-1, 1, size=(B, 2))
    else:
        n = int(name[3:])
        c = NLMPC(OSCILLATORS6 if n == 6 else OSCILLATORS8, 20 if n == 6 else 30, 10 if n == 6 else 15, 0.1)
        c.setOptimizerParameters(NLParameters(maximum_iteration=200))
        x0 = rng.uniform(-0.1, 0.1, size=(B, 2 * n)); x0[:, 0] += 1.0
    return c, torch.from_numpy(x0), torch.zeros(B, c.nu, dtype=torch.float64)


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "ugv"
    B = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 4096
    reps = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 3
    c, x0, u0 = make(name, B)
    b, out = c.make_batch(x0, u0)
    ms = c.time_launches(b, 1)          # warm-up (also grows the workspace)
    ms = c.time_launches(b, reps)
    st = out["solver_status"].cpu().numpy(); it = out["iterations"].cpu().numpy()
    bad = np.concatenate([np.nonzero(st == k)[0][:2] for k in (-1, -3, -4)])
    for i in bad:
        print("failed", int(i), int(st[i]), int(it[i]), x0[i].numpy().tolist(), file=sys.stderr)
    print(json.dumps(dict(workload=name, batch=B, nz=c.nz, ms_per_batch=ms, solves_per_s=B / ms * 1e3,
                          solver_status_counts={int(k): int((st == k).sum()) for k in np.unique(st)},
                          iterations_mean=float(it.mean()), iterations_max=int(it.max()))))
