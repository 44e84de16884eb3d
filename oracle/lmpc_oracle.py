"""TEST INFRASTRUCTURE ONLY -- ctypes loader for the C oracle (liblmpc_oracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

RS_SUCCESS, RS_MAX_ITERATION, RS_INFEASIBLE, RS_ERROR, RS_UNKNOWN = range(5)


class LParams(C.Structure):
    _fields_ = [("maximum_iteration", C.c_int), ("time_limit", C.c_double), ("enable_warm_start", C.c_int),
                ("alpha", C.c_double), ("rho", C.c_double), ("eps_rel", C.c_double), ("eps_abs", C.c_double),
                ("eps_prim_inf", C.c_double), ("eps_dual_inf", C.c_double),
                ("verbose", C.c_int), ("adaptive_rho", C.c_int), ("polish", C.c_int),
                ("adaptive_rho_interval", C.c_int), ("nan_faithful", C.c_int)]


class Result(C.Structure):
    _fields_ = [("solver_status", C.c_int), ("status", C.c_int), ("is_feasible", C.c_int), ("iters", C.c_int),
                ("polished", C.c_int), ("rho_updates", C.c_int), ("cost", C.c_double), ("rho", C.c_double)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liblmpc_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.oracle_lmpc_create.restype = C.c_void_p
        _LIB.oracle_lmpc_solve_batch_constref.restype = C.c_double
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f(a):
    """column-major contiguous float64 copy"""
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def default_params(**kw):
    p = LParams()
    lib().oracle_lparams_default(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class OracleLMPC:
    """Mirrors the reference builder/optimizer pair for one controller."""

    def __init__(self, nx, nu, ndu, ny, ph, ch):
        self.nx, self.nu, self.ndu, self.ny, self.ph, self.ch = nx, nu, ndu, ny, ph, ch
        self.L = lib()
        self.h = C.c_void_p(self.L.oracle_lmpc_create(nx, nu, ndu, ny, ph, ch))
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self.L.oracle_lmpc_sizes(self.h, C.byref(a), C.byref(b), C.byref(c))
        self.nvar, self.ncon, self.neq = a.value, b.value, c.value
        self.params = default_params()

    def __del__(self):
        try:
            self.L.oracle_lmpc_destroy(self.h)
        except Exception:
            pass

    def set_model(self, A, B, Cm):
        A, B, Cm = _f(A), _f(B), _f(Cm)
        return self.L.oracle_lmpc_set_model(self.h, _p(A), _p(B), _p(Cm))

    def set_exogenous(self, Bd, Dd):
        Bd, Dd = _f(Bd), _f(Dd)
        return self.L.oracle_lmpc_set_exogenous(self.h, _p(Bd), _p(Dd))

    def set_objective(self, OW, UW, DUW):
        OW, UW, DUW = _f(OW), _f(UW), _f(DUW)
        return self.L.oracle_lmpc_set_objective(self.h, _p(OW), _p(UW), _p(DUW))

    def set_objective_idx(self, i, ow, uw, duw):
        ow, uw, duw = _f(ow), _f(uw), _f(duw)
        return self.L.oracle_lmpc_set_objective_idx(self.h, int(i), _p(ow), _p(uw), _p(duw))

    def set_state_bounds(self, lo, hi):
        lo, hi = _f(lo), _f(hi)
        return self.L.oracle_lmpc_set_state_bounds(self.h, _p(lo), _p(hi))

    def set_state_bounds_idx(self, i, lo, hi):
        lo, hi = _f(lo), _f(hi)
        return self.L.oracle_lmpc_set_state_bounds_idx(self.h, int(i), _p(lo), _p(hi))

    def set_output_bounds(self, lo, hi):
        lo, hi = _f(lo), _f(hi)
        return self.L.oracle_lmpc_set_output_bounds(self.h, _p(lo), _p(hi))

    def set_output_bounds_idx(self, i, lo, hi):
        lo, hi = _f(lo), _f(hi)
        return self.L.oracle_lmpc_set_output_bounds_idx(self.h, int(i), _p(lo), _p(hi))

    def set_input_bounds(self, lo, hi):
        lo, hi = _f(lo), _f(hi)
        return self.L.oracle_lmpc_set_input_bounds(self.h, _p(lo), _p(hi))

    def set_input_bounds_idx(self, i, lo, hi):
        lo, hi = _f(lo), _f(hi)
        return self.L.oracle_lmpc_set_input_bounds_idx(self.h, int(i), _p(lo), _p(hi))

    def set_scalar(self, smin, smax, X, U):
        smin, smax, X, U = _f(smin), _f(smax), _f(X), _f(U)
        return self.L.oracle_lmpc_set_scalar(self.h, _p(smin), _p(smax), _p(X), _p(U))

    def set_scalar_idx(self, i, smin, smax, X, U):
        X, U = _f(X), _f(U)
        return self.L.oracle_lmpc_set_scalar_idx(self.h, int(i), C.c_double(smin), C.c_double(smax), _p(X), _p(U))

    def get_problem(self, x0, u0, yRef, uRef, duRef, dMeas, dense=False):
        x0, u0, yRef, uRef, duRef, dMeas = map(_f, (x0, u0, yRef, uRef, duRef, dMeas))
        q = np.zeros(self.nvar); l = np.zeros(self.ncon); u = np.zeros(self.ncon)
        P = np.zeros((self.nvar, self.nvar), order="F") if dense else None
        A = np.zeros((self.ncon, self.nvar), order="F") if dense else None
        self.L.oracle_lmpc_get_problem(self.h, _p(x0), _p(u0), _p(yRef), _p(uRef), _p(duRef), _p(dMeas),
                                       _p(P), _p(q), _p(A), _p(l), _p(u))
        return (P, q, A, l, u) if dense else (q, l, u)

    def solve(self, x0, u0, yRef, uRef, duRef, dMeas, want_seq=True):
        x0, u0, yRef, uRef, duRef, dMeas = map(_f, (x0, u0, yRef, uRef, duRef, dMeas))
        ph, nx, ny, nu = self.ph, self.nx, self.ny, self.nu
        res = Result()
        cmd = np.zeros(nu); z = np.zeros(self.nvar); y = np.zeros(self.ncon)
        ss = np.zeros((ph + 1, nx), order="F"); so = np.zeros((ph + 1, ny), order="F"); si = np.zeros((ph + 1, nu), order="F")
        al = np.zeros(self.ncon, dtype=np.uint8); au = np.zeros(self.ncon, dtype=np.uint8)
        self.L.oracle_lmpc_solve(self.h, C.byref(self.params), _p(x0), _p(u0), _p(yRef), _p(uRef), _p(duRef), _p(dMeas),
                                 C.byref(res), _p(cmd), _p(z), _p(y), _p(ss), _p(so), _p(si), _p(al), _p(au))
        return dict(cmd=cmd, cost=res.cost, status=res.status, solver_status=res.solver_status,
                    is_feasible=bool(res.is_feasible), iters=res.iters, polished=res.polished, rho=res.rho,
                    rho_updates=res.rho_updates, z=z, y=y, state=ss, output=so, input=si,
                    active_lower=al.astype(bool), active_upper=au.astype(bool))

    def solve_batch_constref(self, x0, u0, yref, want_active=False):
        x0 = np.ascontiguousarray(x0, dtype=np.float64); u0 = np.ascontiguousarray(u0, dtype=np.float64)
        yref = np.ascontiguousarray(yref, dtype=np.float64)
        B = x0.shape[0]
        cmd = np.zeros((B, self.nu)); cost = np.zeros(B)
        status = np.zeros(B, dtype=np.int32); sst = np.zeros(B, dtype=np.int32)
        iters = np.zeros(B, dtype=np.int32); pol = np.zeros(B, dtype=np.int32)
        al = np.zeros((B, self.ncon), dtype=np.uint8) if want_active else None
        au = np.zeros((B, self.ncon), dtype=np.uint8) if want_active else None
        per = np.zeros(B)
        t = self.L.oracle_lmpc_solve_batch_constref(self.h, C.byref(self.params), B, _p(x0), _p(u0), _p(yref),
                                                    _p(cmd), _p(cost), _p(status), _p(sst), _p(iters), _p(pol),
                                                    _p(al), _p(au), _p(per))
        return dict(cmd=cmd, cost=cost, status=status, solver_status=sst, iters=iters, polished=pol,
                    active_lower=al, active_upper=au, seconds=t, per_solve_seconds=per)
