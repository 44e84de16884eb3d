"""Adds the BOUND-ACTIVE oscillator sets to tests/golden/nlmpc_oracle_solutions.json (the other keys of that file are left as they are):
the reference's networked_oscillators_ex.cpp system with NLMPC::setInputBounds(-ub, +ub) on every step of the control horizon
(NLOptimizer.hpp:346-404), tight enough that most inputs sit on a bound at the optimum -- working sets that fill the sub-problem's
variables, the case in which round 5's inverse form lost instances.

  oscillators6_ph20_ch10_bounds: 16 instances at |u| <= 0.05, 8 at |u| <= 0.15      (the example's own shape)
  oscillators8_ph30_ch15_bounds: 16 instances at |u| <= 0.05, 8 at |u| <= 0.15      (BASELINE config 5's shape)

The oracle is oracle/nlmpc_c.py (the reference's transcription restated in C driving scipy's SLSQP).  Starts: x0[0] = 1 + U(-0.5, 0.5),
the rest U(-0.5, 0.5) -- the distribution of tests/hunt_inconsistent_gpu.py -- default_rng(606 + N).

Run from the repository root (~10 min on 6 cores):  python tests/golden/make_nlmpc_bounds_golden.py"""
import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def _solve(job):
    name, x0, ub = job
    from oracle import nlmpc_c
    m = nlmpc_c.make(name)
    N = m.nu
    o = m.solve(np.asarray(x0), np.zeros(N), max_iter=400, hard=True, lb_u=[-ub] * N, ub_u=[ub] * N)
    g, _ = m.user_ineq(o["z"], False)
    c, _ = m.state_eq(o["z"], False)
    u = o["z"][m.ph * m.nx:m.ph * m.nx + m.ch * N]
    return dict(x0=list(map(float, x0)), u0=[0.0] * N, ub=ub, cmd=o["cmd"].tolist(), cost=o["cost"], success=bool(o["success"]), nit=o["nit"],
                slsqp_mode=o["slsqp_mode"], eq_violation=float(np.abs(c).max()), ineq_violation=float(max(0.0, g.max())),
                inputs_on_a_bound=int((np.abs(np.abs(u) - ub) <= 1e-9).sum()))


def main():
    jobs = []
    for name, N in (("osc6", 6), ("osc8", 8)):
        rng = np.random.default_rng(606 + N)
        X0 = rng.uniform(-0.5, 0.5, size=(24, 2 * N)); X0[:, 0] += 1.0
        jobs += [(name, X0[i], 0.05 if i < 16 else 0.15) for i in range(24)]
    with mp.Pool(max(1, (os.cpu_count() or 2) - 2)) as pool:
        res = pool.map(_solve, jobs, chunksize=1)
    path = os.path.join(ROOT, "tests", "golden", "nlmpc_oracle_solutions.json")
    out = json.load(open(path))
    out["oscillators6_ph20_ch10_bounds"] = dict(model="oscillators", N=6, ph=20, ch=10, Ts=0.1, hard=True, cases=res[:24])
    out["oscillators8_ph30_ch15_bounds"] = dict(model="oscillators", N=8, ph=30, ch=15, Ts=0.1, hard=True, cases=res[24:])
    json.dump(out, open(path, "w"), indent=None, separators=(",", ":"))
    for k in ("oscillators6_ph20_ch10_bounds", "oscillators8_ph30_ch15_bounds"):
        v = out[k]["cases"]
        print(k, len(v), "cases,", sum(c["success"] for c in v), "converged, inputs on a bound per case:", [c["inputs_on_a_bound"] for c in v])


if __name__ == "__main__":
    main()
