"""Generates tests/golden/nlmpc_oracle_solutions.json: optimal first moves / costs of the NLMPC oracle on instances of
BASELINE configs 3 and 5 that take the oracle too long to solve inside the test suite (config 5: 60 ... 90 s per instance).

The oracle here is oracle/nlmpc_c.py -- the reference's transcription restated in C (oracle/nlmpc_callbacks.c) driving scipy's
SLSQP; tests/test_nlmpc_oracle.py pins it against the numpy restatement (oracle/nlmpc_numpy.py), which produced the first two
config-5 cases of the previous file (kept: the new run reproduces them).  The instances are the first ones of bench.py's
synthetic batches (nl_make: default_rng(0)), i.e. what the quoted throughput is measured on, plus the examples' own start.

Run from the repository root (all host cores, ~10 min on 8):  python tests/golden/make_nlmpc_golden.py"""
import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

N_OSC8, N_UGV = 64, 256


def _solve(job):
    name, x0, u0, iters, hard = job
    from oracle import nlmpc_c
    m = nlmpc_c.make(name)
    o = m.solve(np.asarray(x0), np.asarray(u0), max_iter=iters, hard=hard)
    # SLSQP's exit mode 8 ("positive directional derivative for linesearch") at a feasible point is how scipy's port ends most UGV
    # solves: the step has shrunk below what ftol = 1e-12 can see.  The violations are stored so that a test can tell such an end
    # point from a failed solve.
    g, _ = m.user_ineq(o["z"], False)
    c, _ = m.state_eq(o["z"], False)
    return dict(x0=list(map(float, x0)), u0=list(map(float, u0)), cmd=o["cmd"].tolist(), cost=o["cost"], success=bool(o["success"]), nit=o["nit"],
                slsqp_mode=o["slsqp_mode"], eq_violation=float(np.abs(c).max()), ineq_violation=float(max(0.0, g.max())))


def main():
    rng = np.random.default_rng(0)
    X8 = rng.uniform(-0.1, 0.1, size=(1024, 16)); X8[:, 0] += 1.0            # bench.py nl_make("osc8")
    x8 = [np.eye(16)[0]] + [X8[i] for i in range(N_OSC8 - 1)]                 # networked_oscillators_ex.cpp's own start first
    rng = np.random.default_rng(0)
    Xu = np.zeros((4096, 4)); Xu[:, :2] = rng.uniform(-0.5, 0.5, size=(4096, 2))   # bench.py nl_make("ugv")
    xu = [np.zeros(4)] + [Xu[i] for i in range(N_UGV - 1)]                    # ugv_ex.cpp's own start first
    jobs = [("osc8", x, np.zeros(8), 200, True) for x in x8] + [("ugv", x, np.zeros(2), 150, False) for x in xu]
    with mp.Pool(max(1, (os.cpu_count() or 2) - 2)) as pool:
        res = pool.map(_solve, jobs, chunksize=1)
    out = {"oscillators8_ph30_ch15": dict(model="oscillators", N=8, ph=30, ch=15, Ts=0.1, hard=True, cases=res[:N_OSC8]),
           "ugv_ph30_ch30": dict(model="ugv", ph=30, ch=30, Ts=0.1, hard=False, cases=res[N_OSC8:])}
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "nlmpc_oracle_solutions.json"), "w"), indent=None, separators=(",", ":"))
    for k, v in out.items():
        print(k, len(v["cases"]), "cases,", sum(c["success"] for c in v["cases"]), "converged,", sum(c["slsqp_mode"] == 8 for c in v["cases"]), "ended in mode 8")


if __name__ == "__main__":
    main()
