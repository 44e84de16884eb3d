"""TEST / BENCHMARK INFRASTRUCTURE ONLY -- the compiled NLMPC CPU baseline: the reference's callbacks restated in C
(oracle/nlmpc_callbacks.c: Objective.hpp:91-265, Constraints.hpp:211-316, 490-905) driving scipy's SLSQP (Kraft's code,
compiled; NLopt's LD_SLSQP is a translation of it).  This is what bench.py times as `cpu_baseline` for the NLMPC workloads: in
libmpc++ the time of NLOptimizer::run (NLOptimizer.hpp:412-638) goes into exactly these finite-difference callbacks.  The same
formulas in numpy are oracle/nlmpc_numpy.py; tests/test_nlmpc_oracle.py pins the two against each other."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class _Model(C.Structure):
    _fields_ = [("model", C.c_int), ("nx", C.c_int), ("nu", C.c_int), ("ph", C.c_int), ("ch", C.c_int), ("nineq", C.c_int), ("N", C.c_int),
                ("Ts", C.c_double), ("mu", C.c_double), ("k", C.c_double), ("vpx", C.c_double), ("vpy", C.c_double), ("obs", C.c_double * 6)]


def _lib():
    lib = C.CDLL(os.path.join(_HERE, "libnlmpc_callbacks.so"))
    P = C.c_void_p
    lib.nlc_objective.restype = C.c_double
    lib.nlc_objective.argtypes = [P, P, P, P]
    lib.nlc_state_eq.argtypes = [P, P, P, P, P]
    lib.nlc_user_ineq.argtypes = [P, P, P, P, P]
    lib.nlc_nz.argtypes = [P]
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class NlmpcC:
    def __init__(self, name, ph, ch, Ts=0.1, N=0):
        self.lib = _lib()
        m = _Model()
        if name == "vanderpol":
            m.model, m.nx, m.nu, m.nineq = 0, 2, 1, ph + 1
        elif name == "ugv":
            m.model, m.nx, m.nu, m.nineq = 1, 4, 2, 2 * (ph + 1)
            m.vpx = m.vpy = 0.7071067811865476
            m.obs = (C.c_double * 6)(2.0, 1.0, 0.3, 1.0, 1.0, 0.3)
        else:
            m.model, m.nx, m.nu, m.nineq, m.N = 2, 2 * N, N, N * (ph + 1), N
            m.mu, m.k = 1.0, 0.1
        m.ph, m.ch, m.Ts = ph, ch, Ts
        self.m = m
        self.nx, self.nu, self.ph, self.ch, self.nineq = m.nx, m.nu, ph, ch, m.nineq
        self.nz = ph * m.nx + ch * m.nu + 1
        self.x0 = np.zeros(m.nx)

    def objective(self, z, want_grad=True):
        z = np.ascontiguousarray(z, float)
        g = np.zeros(self.nz) if want_grad else None
        f = self.lib.nlc_objective(C.byref(self.m), _p(z), _p(self.x0), _p(g))
        return f, g

    def state_eq(self, z, want_jac=True):
        z = np.ascontiguousarray(z, float)
        c = np.zeros(self.ph * self.nx)
        J = np.zeros((self.ph * self.nx, self.nz)) if want_jac else None
        self.lib.nlc_state_eq(C.byref(self.m), _p(z), _p(self.x0), _p(c), _p(J))
        return c, J

    def user_ineq(self, z, want_jac=True):
        z = np.ascontiguousarray(z, float)
        g = np.zeros(self.nineq)
        J = np.zeros((self.nineq, self.nz)) if want_jac else None
        self.lib.nlc_user_ineq(C.byref(self.m), _p(z), _p(self.x0), _p(g), _p(J))
        return g, J

    def solve(self, x0, u0, max_iter=100, hard=True, lb_u=None, ub_u=None):
        """NLOptimizer::run, cold start (NLOptimizer.hpp:412-638), with scipy's SLSQP; lb_u / ub_u: NLMPC::setInputBounds on every
        step of the control horizon (NLOptimizer.hpp:346-404: bounds on the decision vector)"""
        from scipy.optimize import minimize
        self.x0 = np.ascontiguousarray(x0, float)
        nx, nu, ph, ch = self.nx, self.nu, self.ph, self.ch
        z0 = np.concatenate([np.tile(self.x0, ph), np.tile(np.asarray(u0, float), ch), [0.0]])
        lo = np.full(self.nz, -np.inf); hi = np.full(self.nz, np.inf)
        if lb_u is not None:
            lo[ph * nx:ph * nx + ch * nu] = np.tile(lb_u, ch); hi[ph * nx:ph * nx + ch * nu] = np.tile(ub_u, ch)
        if hard:
            lo[-1] = hi[-1] = 0.0
        cons = [{"type": "eq", "fun": lambda z: self.state_eq(z, False)[0], "jac": lambda z: self.state_eq(z, True)[1]},
                {"type": "ineq", "fun": lambda z: -self.user_ineq(z, False)[0], "jac": lambda z: -self.user_ineq(z, True)[1]}]
        r = minimize(lambda z: self.objective(z, False)[0], z0, jac=lambda z: self.objective(z, True)[1], method="SLSQP",
                     bounds=list(zip(lo, hi)), constraints=cons, options={"maxiter": max_iter, "ftol": 1e-12})
        return dict(z=r.x, cmd=r.x[ph * nx:ph * nx + nu].copy(), cost=float(r.fun), nit=int(r.nit), success=bool(r.success), slsqp_mode=int(r.status))


def make(name):
    """the benchmark workloads of bench.py"""
    return dict(ugv=lambda: NlmpcC("ugv", 30, 30), vanderpol=lambda: NlmpcC("vanderpol", 10, 5, 0.1), osc6=lambda: NlmpcC("osc", 20, 10, 0.1, 6),
                osc8=lambda: NlmpcC("osc", 30, 15, 0.1, 8))[name]()
