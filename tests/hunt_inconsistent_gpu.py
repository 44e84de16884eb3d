"""TEST INFRASTRUCTURE (a script, not collected by pytest): the hunt for the case DESIGN.md section 9 lists as open -- a FEASIBLE non-linear problem at
which the SQP kernels find the linearised constraints inconsistent (solver status -1), the case Kraft's SLSQP relaxes with an auxiliary
variable (NLOptimizer.hpp:519 via nlopt).  Round 5: on the GPU, thousands of instances, on the non-convex rows where a feasible problem can
have an inconsistent linearisation:
  * the UGV (ugv_ex.cpp) with HARD constraints, every instance its own two obstacles (radius 0.3 .. 0.9, placed on and next to the straight
    line of travel), starts a hair outside an obstacle or far from both, input bounds that leave little authority;
  * six oscillators with tight input bounds.
Every instance the kernel ends with -1 is handed to the oracle's SLSQP (scipy, the restated callbacks); a HIT is one the oracle solves to a
feasible point.  Usage: python tests/hunt_inconsistent_gpu.py [instances per setting]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nlmpc_numpy as ref  # noqa: E402
from libmpc_amd.nlmpc import NLMPC, NLParameters, UGV, OSCILLATORS6  # noqa: E402


def ugv_oracle(ph, ch, obs, x0, ub):
    m = ref.ugv(ph=ph, ch=ch)
    o = np.asarray(obs, float).reshape(2, 3)

    def ineq(X, Y, U, e):
        g = np.zeros((X.shape[0], 2))
        for k in range(2):
            g[:, k] = o[k, 2] - np.sqrt((X[:, 0] - o[k, 0]) ** 2 + (X[:, 1] - o[k, 1]) ** 2)
        return g.reshape(-1)
    m.ineq_fun = ineq
    return m.solve(x0, np.zeros(2), max_iter=400, hard=True, lb_u=[-ub, -ub], ub_u=[ub, ub])


n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(2025)
total = ended = hits = oracle_failed = 0
for ph, ch, ub in [(30, 30, 3.0), (30, 30, 1.0), (20, 10, 2.0), (12, 4, 1.5)]:
    c = NLMPC(UGV, ph, ch, 0.1)
    c.setOptimizerParameters(NLParameters(maximum_iteration=200, hard_constraints=1))
    assert c.setInputBounds([-ub, -ub], [ub, ub], (0, ch))
    P = np.zeros((n, 9)); X0 = np.zeros((n, 4))
    for b in range(n):
        # travel along (1, 1) / sqrt 2 at unit speed: the path of the horizon is about ph * 0.1 long
        L = 0.1 * ph
        d = np.array([1.0, 1.0]) / np.sqrt(2.0)
        obs = []
        for k in range(2):
            s = rng.uniform(0.15, 0.9) * L; off = rng.normal(0.0, 0.25); r = rng.uniform(0.3, 0.9)
            cpos = s * d + off * np.array([-d[1], d[0]])
            obs += [cpos[0], cpos[1], r]
        P[b] = [d[0], d[1]] + obs + [0.1]
        # a start just outside the first obstacle (tight margin), or at the origin if that is free
        if rng.uniform() < 0.6:
            ang = rng.uniform(np.pi, 1.5 * np.pi)
            X0[b, :2] = np.array(obs[0:2]) + (obs[2] + 10 ** rng.uniform(-4, -1)) * np.array([np.cos(ang), np.sin(ang)])
        for k in range(2):
            if np.hypot(X0[b, 0] - obs[3 * k], X0[b, 1] - obs[3 * k + 1]) <= obs[3 * k + 2]:
                X0[b, :2] = np.array(obs[3 * k:3 * k + 2]) - (obs[3 * k + 2] + 1e-3) * d
        X0[b, 2:] = rng.uniform(0.0, 1.0, 2)
    bad0 = np.array([max(P[b, 4] - np.hypot(X0[b, 0] - P[b, 2], X0[b, 1] - P[b, 3]), P[b, 7] - np.hypot(X0[b, 0] - P[b, 5], X0[b, 1] - P[b, 6])) > 0 for b in range(n)])
    r = c.optimizeBatch(torch.from_numpy(X0), torch.zeros(n, 2, dtype=torch.float64), params=torch.from_numpy(P))
    torch.cuda.synchronize()
    st = r["solver_status"].cpu().numpy()
    cand = np.nonzero((st == -1) & ~bad0)[0]
    total += n; ended += len(cand)
    print("ugv ph %d ch %d |u| <= %.1f: %d instances, %d solved, %d ended with -1 (%d more from a start inside an obstacle), other codes %s"
          % (ph, ch, ub, n, int((st > 0).sum()), len(cand), int(((st == -1) & bad0).sum()), {int(k): int((st == k).sum()) for k in np.unique(st) if k not in (-1, 3, 4)}), flush=True)
    for b in cand[:24]:
        o = ugv_oracle(ph, ch, P[b, 2:8], X0[b], ub)
        ok = bool(o["success"])
        if ok:
            hits += 1
            print("  HIT: instance %d: the oracle solves it (cost %.6g); obstacles %s start %s" % (b, o["cost"], P[b, 2:8].tolist(), X0[b].tolist()), flush=True)
        else:
            oracle_failed += 1
# six oscillators, tight input bounds
for ub in (0.05, 0.15):
    c = NLMPC(OSCILLATORS6, 20, 10, 0.1)
    c.setOptimizerParameters(NLParameters(maximum_iteration=200))
    assert c.setInputBounds([-ub] * 6, [ub] * 6, (0, 10))
    X0 = rng.uniform(-0.5, 0.5, size=(n, 12)); X0[:, 0] += 1.0
    r = c.optimizeBatch(torch.from_numpy(X0), torch.zeros(n, 6, dtype=torch.float64)); torch.cuda.synchronize()
    st = r["solver_status"].cpu().numpy()
    cand = np.nonzero(st == -1)[0]
    total += n; ended += len(cand)
    print("osc6 |u| <= %.2f: %d instances, %d solved, %d ended with -1, other codes %s"
          % (ub, n, int((st > 0).sum()), len(cand), {int(k): int((st == k).sum()) for k in np.unique(st) if k not in (-1, 3, 4)}), flush=True)
    m = ref.oscillators(N=6, ph=20, ch=10, Ts=0.1)
    for b in cand[:8]:
        o = m.solve(X0[b], np.zeros(6), max_iter=400, lb_u=[-ub] * 6, ub_u=[ub] * 6)
        if o["success"]:
            hits += 1
            print("  HIT: osc6 instance %d" % b, flush=True)
        else:
            oracle_failed += 1
print("%d instances, %d ended with solver status -1 from a feasible start, %d of them checked against the oracle: %d hits, %d where the oracle fails too"
      % (total, ended, hits + oracle_failed, hits, oracle_failed))
