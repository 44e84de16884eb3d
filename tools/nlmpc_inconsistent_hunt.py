"""Looks for the case DESIGN.md section 9 lists as open: a FEASIBLE non-linear problem at which the SQP kernels find the linearised constraints
inconsistent (solver status -1) -- the case Kraft's SLSQP handles with an augmented, relaxed sub-problem (NLOptimizer.hpp:519 via nlopt).
Runs on the CPU: the kernels' source through the emulator (tests/emu), the oracle's SLSQP beside it.  Van der Pol (ph = 10, ch = 5) with
input bounds, a bound on the first state from stage `xs` on, random starts inside that bound; and Van der Pol with the terminal equality.
Every instance the kernel ends with -1 is handed to the oracle: a hit is one that the oracle solves.
Round 4 (profiles/r04_inconsistent_hunt.txt): 13 + 1 settings x 16 starts, 110 instances ended with -1, the oracle fails on every one of them (SLSQP's modes 4 and 8: the problems
themselves are infeasible) -- no hit; the kernels report what NLOptimizer::run reports there.  Usage: python tools/nlmpc_inconsistent_hunt.py [trials]"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nlmpc_numpy as ref  # noqa: E402

EMU = os.path.join(ROOT, "tests", "emu")
exe = os.path.join(tempfile.mkdtemp(), "run_nlmpc")
subprocess.run(["g++", "-O1", "-std=c++20", "-DHIPEMU_WITH_WG", "-I" + EMU, "-I" + os.path.join(ROOT, "include"), "-fpermissive", "-w", "-o", exe,
                os.path.join(EMU, "run_nlmpc.cpp"), os.path.join(EMU, "hipemu_switch.S")], check=True)


def run(args, inst):
    inp = "\n".join(" ".join(repr(float(x)) for x in row) for row in inst) + "\n"
    r = subprocess.run([exe] + [str(a) for a in args], input=inp, capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0, r.stderr[:2000]
    return [json.loads(ln) for ln in r.stdout.splitlines()]


def oracle(m, x0, lo, hi):
    """NlmpcRef.solve with bounds given per entry of z"""
    from scipy.optimize import minimize
    m.x0 = np.asarray(x0, float)
    z0 = np.concatenate([np.tile(m.x0, m.ph), np.zeros(m.ch * m.nu), [0.0]])
    lo = lo.copy(); hi = hi.copy(); lo[-1] = hi[-1] = 0.0
    cons = [{"type": "eq", "fun": lambda z: m.state_eq(z, False)[0], "jac": lambda z: m.state_eq(z, True)[1]}]
    if m.ineq_fun is not None:
        cons.append({"type": "ineq", "fun": lambda z: -m.user_ineq(z)[0], "jac": lambda z: -m.user_ineq(z)[1]})
    if m.eq_fun is not None:
        cons.append({"type": "eq", "fun": lambda z: m.user_eq(z)[0], "jac": lambda z: m.user_eq(z)[1]})
    r = minimize(lambda z: m.objective(z, False)[0], z0, jac=lambda z: m.objective(z, True)[1], method="SLSQP", bounds=list(zip(lo, hi)),
                 constraints=cons, options={"maxiter": 1000, "ftol": 1e-12})
    return bool(r.success), str(r.message)


trials = int(sys.argv[1]) if len(sys.argv) > 1 else 13
rng = np.random.default_rng(11)
ph, ch, B = 10, 5, 16
m = ref.vanderpol(ph=ph, ch=ch, Ts=0.1)
hits = ended = 0
for trial in range(trials):
    bx = rng.uniform(0.2, 0.8); ub = rng.uniform(0.3, 0.5); xs = int(rng.integers(0, 8))
    X0 = np.c_[rng.uniform(-bx, bx, B) * 0.95, rng.uniform(-2, 2, B)]
    r = run(["vanderpol", ph, ch, 0.1, 1, 200, "wave", "lbu=%r" % (-ub), "ubu=%r" % ub, "lbx0=%r" % (-bx), "ubx0=%r" % bx, "xs=%d" % xs], np.hstack([X0, np.zeros((B, 1))]))
    lo = np.full(m.nz, -np.inf); hi = np.full(m.nz, np.inf)
    for i in range(xs, ph):
        lo[i * 2] = -bx; hi[i * 2] = bx
    lo[ph * 2:ph * 2 + ch] = -ub; hi[ph * 2:ph * 2 + ch] = ub
    n = 0
    for b, y in enumerate(r):
        if y["solver_status"] == -1:
            n += 1
            ok, msg = oracle(m, X0[b], lo, hi)
            hits += int(ok)
            if ok:
                print("HIT: x0 %s |x_0| <= %.3f from stage %d, |u| <= %.3f" % (X0[b].tolist(), bx, xs, ub), flush=True)
    ended += n
    print("setting %2d: |x_0| <= %.2f from stage %d, |u| <= %.2f: %2d of %d end with -1" % (trial, bx, xs, ub, n, B), flush=True)
mt = ref.vanderpol_terminal(ph=ph, ch=ch, Ts=0.1)
X0 = rng.uniform(-0.3, 0.3, (B, 2))
r = run(["vanderpol_terminal", ph, ch, 0.1, 1, 300, "wave"], np.hstack([X0, np.zeros((B, 1))]))
n = 0
for b, y in enumerate(r):
    if y["solver_status"] == -1:
        n += 1
        ok, msg = oracle(mt, X0[b], np.full(mt.nz, -np.inf), np.full(mt.nz, np.inf))
        hits += int(ok)
ended += n
print("terminal equality: %d of %d end with -1" % (n, B))
print("%d instances ended with -1; the oracle solves %d of them" % (ended, hits))
