"""TEST INFRASTRUCTURE ONLY -- dense numpy restatement of the OSQP 0.6.3 algorithm.

The reference calls OSQP v0.6.3 (configure.sh:39-42) from LOptimizer::run
(include/mpc/LMPC/LOptimizer.hpp:244-284).  OSQP's sources are not in the
reference tree; this file restates the *published* algorithm (Stellato et al.,
"OSQP: an operator splitting solver for quadratic programs", cited at
docs/source/cite/cite.rst:60-70) with the v0.6.3 default settings and the
overrides libmpc++ applies (LOptimizer.hpp:246-257, Types.hpp:146-161).
Dense linear algebra (scipy LDL^T via numpy solves) replaces QDLDL: same
mathematics, different round-off.

PARITY NOTE: OSQP picks its adaptive-rho interval from wall-clock time
(adaptive_rho_interval=0 with a PROFILING build); this restatement fixes it at
25 iterations (= check_termination, the value the time rule rounds to whenever
0.4 x setup-time is shorter than ~37 iterations).  ADMM iterates are therefore
not reproducible against a real OSQP run; the polished solution is.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla

OSQP_INFTY = 1e30
MIN_SCALING = 1e-4
MAX_SCALING = 1e4
RHO_MIN = 1e-6
RHO_MAX = 1e6
RHO_EQ_OVER_RHO_INEQ = 1e3
RHO_TOL = 1e-4
DIV_TOL = 1.0 / OSQP_INFTY

# status values (osqp constants.h, v0.6.3)
OSQP_DUAL_INFEASIBLE_INACCURATE = 4
OSQP_PRIMAL_INFEASIBLE_INACCURATE = 3
OSQP_SOLVED_INACCURATE = 2
OSQP_SOLVED = 1
OSQP_MAX_ITER_REACHED = -2
OSQP_PRIMAL_INFEASIBLE = -3
OSQP_DUAL_INFEASIBLE = -4
OSQP_NON_CVX = -7
OSQP_UNSOLVED = -10


class Settings:
    def __init__(self, **kw):
        # libmpc++ LParameters defaults (Types.hpp:108-114,150-160)
        self.alpha = 1.6
        self.rho = 1e-6
        self.eps_rel = 1e-4
        self.eps_abs = 1e-4
        self.eps_prim_inf = 1e-3
        self.eps_dual_inf = 1e-3
        self.max_iter = 100
        self.adaptive_rho = True
        self.polish = True
        # OSQP v0.6.3 defaults the reference inherits
        self.sigma = 1e-6
        self.scaling = 10
        self.adaptive_rho_interval = 25     # see PARITY NOTE
        self.adaptive_rho_tolerance = 5.0
        self.check_termination = 25
        self.delta = 1e-6
        self.polish_refine_iter = 3
        for k, v in kw.items():
            setattr(self, k, v)


def _limit(v):
    v = np.where(v < MIN_SCALING, 1.0, v)
    return np.minimum(v, MAX_SCALING)


def _ruiz(P, q, A, l, u, iters):
    n, m = P.shape[0], A.shape[0]
    D = np.ones(n); E = np.ones(m); c = 1.0
    P = P.copy(); q = q.copy(); A = A.copy()
    for _ in range(iters):
        dn = np.maximum(np.abs(P).max(axis=0), np.abs(A).max(axis=0) if m else 0.0)
        en = np.abs(A).max(axis=1) if m else np.zeros(0)
        dt = 1.0 / np.sqrt(_limit(dn))
        et = 1.0 / np.sqrt(_limit(en))
        P = dt[:, None] * P * dt[None, :]
        A = et[:, None] * A * dt[None, :]
        q = dt * q
        D *= dt; E *= et
        cn = np.abs(P).max(axis=0).mean()
        qn = np.abs(q).max()
        qn = 1.0 if qn < MIN_SCALING else min(qn, MAX_SCALING)
        ct = max(cn, qn)
        ct = 1.0 if ct < MIN_SCALING else min(ct, MAX_SCALING)
        ct = 1.0 / ct
        P *= ct; q *= ct; c *= ct
    with np.errstate(invalid="ignore"):
        l = E * l; u = E * u
    return P, q, A, l, u, D, E, c


class _Kkt:
    def __init__(self, P, A, sigma, rho_vec):
        n, m = P.shape[0], A.shape[0]
        K = np.zeros((n + m, n + m))
        K[:n, :n] = P + sigma * np.eye(n)
        K[:n, n:] = A.T
        K[n:, :n] = A
        K[n:, n:] = -np.diag(1.0 / rho_vec)
        self.lu = sla.lu_factor(K)

    def solve(self, rhs):
        return sla.lu_solve(self.lu, rhs)


def solve(P, q, A, l, u, s: Settings, x0=None, y0=None):
    """Returns dict(x, y, status, iters, obj, polished, active_lower, active_upper)."""
    n, m = P.shape[0], A.shape[0]
    Ps, qs, As, ls, us, D, E, c = _ruiz(P, q, A, l, u, s.scaling) if s.scaling else \
        (P, q, A, l, u, np.ones(n), np.ones(m), 1.0)
    Dinv, Einv, cinv = 1.0 / D, 1.0 / E, 1.0 / c

    loose = (ls < -OSQP_INFTY * MIN_SCALING) & (us > OSQP_INFTY * MIN_SCALING)
    with np.errstate(invalid="ignore"):
        eq = (us - ls) < RHO_TOL
    ctype = np.where(loose, -1, np.where(eq, 1, 0))

    def make_rho(rho):
        return np.where(ctype == -1, RHO_MIN, np.where(ctype == 1, RHO_EQ_OVER_RHO_INEQ * rho, rho))

    rho = min(max(s.rho, RHO_MIN), RHO_MAX)
    rho_vec = make_rho(rho)
    kkt = _Kkt(Ps, As, s.sigma, rho_vec)

    x = np.zeros(n); z = np.zeros(m); y = np.zeros(m)
    if x0 is not None:
        x = Dinv * x0
        z = As @ x
        y = c * Einv * y0
    lsc = np.where(ls < -OSQP_INFTY * MIN_SCALING, -np.inf, ls)
    usc = np.where(us > OSQP_INFTY * MIN_SCALING, np.inf, us)

    status = OSQP_UNSOLVED
    delta_x = np.zeros(n); delta_y = np.zeros(m)
    rho_updates = 0

    def residuals():
        Ax = As @ x
        Px = Ps @ x
        Aty = As.T @ y
        pri = np.abs(Einv * (Ax - z)).max() if m else 0.0
        dua = cinv * np.abs(Dinv * (Px + qs + Aty)).max()
        eps_pri = s.eps_abs + s.eps_rel * max(np.abs(Einv * z).max(), np.abs(Einv * Ax).max()) if m else s.eps_abs
        eps_dua = s.eps_abs + s.eps_rel * cinv * max(np.abs(Dinv * qs).max(), np.abs(Dinv * Aty).max(),
                                                       np.abs(Dinv * Px).max())
        return pri, dua, eps_pri, eps_dua

    def primal_infeasible():
        dy = delta_y.copy()
        inf_u = us > OSQP_INFTY * MIN_SCALING
        inf_l = ls < -OSQP_INFTY * MIN_SCALING
        dy[inf_u & inf_l] = 0.0
        only_u = inf_u & ~inf_l
        dy[only_u] = np.minimum(dy[only_u], 0.0)
        only_l = inf_l & ~inf_u
        dy[only_l] = np.maximum(dy[only_l], 0.0)
        nrm = np.abs(E * dy).max() if m else 0.0
        if nrm > DIV_TOL:
            lhs = 0.0
            pos = np.maximum(dy, 0); neg = np.minimum(dy, 0)
            lhs = np.sum(np.where(pos != 0, us * pos, 0.0)) + np.sum(np.where(neg != 0, ls * neg, 0.0))
            if lhs < -s.eps_prim_inf * nrm:
                return np.abs(Dinv * (As.T @ dy)).max() < s.eps_prim_inf * nrm
        return False

    def dual_infeasible():
        nrm = np.abs(D * delta_x).max()
        cost_s = c
        if nrm > DIV_TOL:
            if qs @ delta_x < -cost_s * s.eps_dual_inf * nrm:
                if np.abs(Dinv * (Ps @ delta_x)).max() < cost_s * s.eps_dual_inf * nrm:
                    Adx = Einv * (As @ delta_x)
                    for i in range(m):
                        if ((us[i] < OSQP_INFTY * MIN_SCALING and Adx[i] > s.eps_dual_inf * nrm) or
                                (ls[i] > -OSQP_INFTY * MIN_SCALING and Adx[i] < -s.eps_dual_inf * nrm)):
                            return False
                    return True
        return False

    def check(approx):
        nonlocal status
        pri, dua, eps_pri, eps_dua = residuals()
        ea, er, epi, edi = s.eps_abs, s.eps_rel, s.eps_prim_inf, s.eps_dual_inf
        if pri > OSQP_INFTY or dua > OSQP_INFTY:
            status = OSQP_NON_CVX
            return True
        if approx:
            eps_pri = 10 * (eps_pri); eps_dua = 10 * eps_dua
            epi *= 10; edi *= 10
        pri_ok = (m == 0) or pri < eps_pri
        prim_inf = False if pri_ok else _pi(epi)
        dua_ok = dua < eps_dua
        dual_inf = False if dua_ok else _di(edi)
        if pri_ok and dua_ok:
            status = OSQP_SOLVED_INACCURATE if approx else OSQP_SOLVED
            return True
        if prim_inf:
            status = OSQP_PRIMAL_INFEASIBLE_INACCURATE if approx else OSQP_PRIMAL_INFEASIBLE
            return True
        if dual_inf:
            status = OSQP_DUAL_INFEASIBLE_INACCURATE if approx else OSQP_DUAL_INFEASIBLE
            return True
        return False

    def _pi(eps):
        old = s.eps_prim_inf
        s.eps_prim_inf = eps
        r = primal_infeasible()
        s.eps_prim_inf = old
        return r

    def _di(eps):
        old = s.eps_dual_inf
        s.eps_dual_inf = eps
        r = dual_infeasible()
        s.eps_dual_inf = old
        return r

    it = 0
    done = False
    for it in range(1, s.max_iter + 1):
        x_prev = x.copy(); z_prev = z.copy()
        rhs = np.concatenate([s.sigma * x_prev - qs, z_prev - y / rho_vec])
        sol = kkt.solve(rhs)
        xt = sol[:n]
        zt = z_prev + (sol[n:] - y) / rho_vec
        x = s.alpha * xt + (1 - s.alpha) * x_prev
        delta_x = x - x_prev
        zr = s.alpha * zt + (1 - s.alpha) * z_prev
        z = np.clip(zr + y / rho_vec, lsc, usc)
        delta_y = rho_vec * (zr - z)
        y = y + delta_y
        can_check = s.check_termination and (it % s.check_termination == 0)
        if can_check and check(False):
            done = True
            break
        if s.adaptive_rho and s.adaptive_rho_interval and it % s.adaptive_rho_interval == 0:
            Ax = As @ x
            pr = np.abs(Ax - z).max() if m else 0.0
            pn = max(np.abs(z).max(), np.abs(Ax).max()) if m else 0.0
            Px = Ps @ x; Aty = As.T @ y
            dr = np.abs(Px + qs + Aty).max()
            dn = max(np.abs(qs).max(), np.abs(Aty).max(), np.abs(Px).max())
            pr /= (pn + DIV_TOL); dr /= (dn + DIV_TOL)
            rho_new = min(max(rho * np.sqrt(pr / dr), RHO_MIN), RHO_MAX)
            if rho_new > rho * s.adaptive_rho_tolerance or rho_new < rho / s.adaptive_rho_tolerance:
                rho = rho_new
                rho_vec = make_rho(rho)
                kkt = _Kkt(Ps, As, s.sigma, rho_vec)
                rho_updates += 1
    if not done:
        can_check = s.check_termination and (it % s.check_termination == 0)
        if not can_check:
            check(False)
        if status == OSQP_UNSOLVED:
            if not check(True):
                status = OSQP_MAX_ITER_REACHED

    polished = 0
    act_lo = np.zeros(m, bool); act_up = np.zeros(m, bool)
    if s.polish and status == OSQP_SOLVED:
        with np.errstate(invalid="ignore"):
            act_lo = (z - ls) < -y
            act_up = (us - z) < y
        il = np.nonzero(act_lo)[0]; iu = np.nonzero(act_up)[0]
        Ared = np.vstack([As[il], As[iu]])
        na_ = Ared.shape[0]
        K = np.zeros((n + na_, n + na_))
        K[:n, :n] = Ps; K[:n, n:] = Ared.T; K[n:, :n] = Ared
        Kreg = K.copy()
        Kreg[:n, :n] += s.delta * np.eye(n)
        Kreg[n:, n:] -= s.delta * np.eye(na_)
        lu = sla.lu_factor(Kreg)
        rhs = np.concatenate([-qs, ls[il], us[iu]])
        sol = sla.lu_solve(lu, rhs)
        for _ in range(s.polish_refine_iter):
            sol = sol + sla.lu_solve(lu, rhs - K @ sol)
        xp = sol[:n]
        zp = np.clip(As @ xp, lsc, usc)
        yp = np.zeros(m)
        yp[iu] = sol[n + len(il):]
        yp[il] = sol[n:n + len(il)]
        pri0, dua0, _, _ = residuals()
        xs, zs, ys = x, z, y
        x, z, y = xp, zp, yp
        pri1, dua1, _, _ = residuals()
        ok = (pri1 < pri0 and dua1 < dua0) or (pri1 < pri0 and dua0 < 1e-10) or (dua1 < dua0 and pri0 < 1e-10)
        if ok:
            polished = 1
        else:
            polished = -1
            x, z, y = xs, zs, ys
    xo = D * x
    yo = cinv * E * y
    obj = 0.5 * xo @ P @ xo + q @ xo
    return dict(x=xo, y=yo, status=status, iters=it, obj=obj, polished=polished,
                active_lower=act_lo, active_upper=act_up, rho_updates=rho_updates, rho=rho)
