"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's non-linear MPC transcription.

Restates what NLopt is handed by libmpc++ on every SLSQP evaluation:
  Mapping      include/mpc/NLMPC/Mapping.hpp:174-257   decision vector <-> (X, U, slack), move blocking
  Objective    include/mpc/NLMPC/Objective.hpp:91-265   user cost + forward-difference gradient
  Constraints  include/mpc/NLMPC/Constraints.hpp:490-905 dynamics equalities (trapezoidal collocation /
               one-step) + central-difference block Jacobians, user inequalities + Jacobians
with the reference's quirks: the finite-difference step of the whole-horizon
functions is taken from `Xa.array()(j)` = element (row j, column 0) of the (ph+1) x n matrix, not from
the perturbed element; the last input row is perturbed together with its copy; inequality Jacobian
columns are multiplied by the state scaling, the objective gradient is not.

The optimiser itself (NLopt LD_SLSQP, an unpinned master checkout, configure.sh:26) is not in the
reference tree.  solve() drives scipy's SLSQP -- Kraft's original code, which NLopt's is a translation
of -- with these callbacks.  PARITY STATUS: the components are pinned by the reference's known answers
(tests/test_nlmpc_oracle.py); the end-to-end NLMPC solve is UNPINNED (no reference test asserts one).
"""
from __future__ import annotations

import numpy as np

DV = np.sqrt(np.finfo(float).eps)          # Objective.hpp:283, Constraints.hpp (same constant)


class NlmpcRef:
    def __init__(self, nx, nu, ny, ph, ch, ineq, eq=0):
        self.nx, self.nu, self.ny, self.ph, self.ch, self.ineq, self.eq = nx, nu, ny, ph, ch, ineq, eq
        self.nz = ph * nx + ch * nu + 1                     # Objective.hpp:45
        self.input_scaling = np.ones(nu)
        self.state_scaling = np.ones(nx)
        self.continuous = False
        self.Ts = 0.0
        self.f = None          # f(x, u, step) -> dx or x+
        self.out = None        # out(x, u, step) -> y
        self.cost = None       # cost(X, Y, U, e) -> float
        self.ineq_fun = None   # ineq_fun(X, Y, U, e) -> [ineq]
        self.eq_fun = None     # eq_fun(X, U) -> [eq]
        self.x0 = np.zeros(nx)
        self._mapping()

    # -- Mapping::computeMapping (Mapping.hpp:221-257) ------------------------------------------
    def _mapping(self):
        nu, ph, ch = self.nu, self.ph, self.ch
        m = np.ones(ch, dtype=int); m[ch - 1] = ph - ch + 1
        self.Iz2u = np.zeros((ph * nu, ch * nu))
        self.Iu2z = np.zeros((ch * nu, ph * nu))
        ix = jx = 0
        for i in range(ch):
            self.Iu2z[ix:ix + nu, jx:jx + nu] = np.diag(1.0 / self.input_scaling)
            for _ in range(m[i]):
                self.Iz2u[jx:jx + nu, ix:ix + nu] = np.diag(self.input_scaling)
                jx += nu
            ix += nu

    def set_scaling(self, input_scaling=None, state_scaling=None):
        if input_scaling is not None:
            self.input_scaling = np.asarray(input_scaling, float)
        if state_scaling is not None:
            self.state_scaling = np.asarray(state_scaling, float)
        self._mapping()

    # -- Mapping::unwrapVector (Mapping.hpp:174-211) --------------------------------------------
    def unwrap(self, z):
        nx, nu, ph, ch = self.nx, self.nu, self.ph, self.ch
        z = np.asarray(z, float)
        U = np.zeros((ph + 1, nu))
        U[:ph] = (self.Iz2u @ z[ph * nx: ph * nx + ch * nu]).reshape(ph, nu)
        U[ph] = U[ph - 1]
        X = np.zeros((ph + 1, nx))
        X[0] = self.x0
        X[1:] = z[:ph * nx].reshape(ph, nx)
        X = X / self.state_scaling[None, :]          # every row, x0 included (:204)
        return X, U, z[-1]

    # -- Model::getOutput (Model.hpp:72-96) -----------------------------------------------------
    def outputs(self, X, U):
        Y = np.zeros((self.ph + 1, self.ny))
        if self.out is not None:
            for i in range(self.ph + 1):
                Y[i] = self.out(X[i], U[i], i)
        return Y

    # -- Objective::evaluate + computeGradient (Objective.hpp:91-265) ---------------------------
    def objective(self, z, want_grad=True):
        nx, nu, ph, ch = self.nx, self.nu, self.ph, self.ch
        X, U, e = self.unwrap(z)
        fu = lambda X_, U_, e_: float(self.cost(X_, self.outputs(X_, U_), U_, e_))
        f0 = fu(X, U, e)
        if not want_grad:
            return f0, None
        Jx = np.zeros((nx, ph)); Jmv = np.zeros((nu, ph))
        Xa = np.maximum(np.abs(X), 1.0); Ua = np.maximum(np.abs(U), 1.0)
        lin = lambda M, j: M.reshape(-1, order="F")[j]           # Eigen .array()(j): column-major linear index
        Xp = X.copy(); Up = U.copy()
        for i in range(ph):
            for j in range(nx):
                dx = DV * lin(Xa, j)
                Xp[i + 1, j] += dx
                Jx[j, i] = (fu(Xp, Up, e) - f0) / dx
                Xp[i + 1, j] -= dx
        for i in range(ph - 1):
            for j in range(nu):
                du = DV * lin(Ua, j)
                Up[i, j] += du
                Jmv[j, i] = (fu(Xp, Up, e) - f0) / du
                Up[i, j] -= du
        for j in range(nu):
            du = DV * lin(Ua, j)
            Up[ph - 1, j] += du; Up[ph, j] += du
            Jmv[j, ph - 1] = (fu(Xp, Up, e) - f0) / du
            Up[ph - 1, j] -= du; Up[ph, j] -= du
        de = max(DV, abs(e)) * DV
        Je = (fu(Xp, Up, e + de) - fu(Xp, Up, e - de)) / (2 * de)
        g = np.concatenate([Jx.reshape(-1, order="F"), self.Iz2u.T @ Jmv.reshape(-1, order="F"), [Je]])
        return f0, g

    # -- Constraints::computeStateEqJacobian (Constraints.hpp:844-905) ---------------------------
    def _state_jac(self, x, u, p):
        nx, nu = self.nx, self.nu
        A = np.zeros((nx, nx)); B = np.zeros((nx, nu))
        Xa = np.maximum(np.abs(x), 1.0); Ua = np.maximum(np.abs(u), 1.0)
        for i in range(nx):
            dx = DV * Xa[i]
            xp = x.copy(); xm = x.copy(); xp[i] += dx; xm[i] -= dx
            A[:, i] = (np.asarray(self.f(xp, u, p)) - np.asarray(self.f(xm, u, p))) / (2 * dx)
        for i in range(nu):
            du = DV * Ua[i]
            up = u.copy(); um = u.copy(); up[i] += du; um[i] -= du
            B[:, i] = (np.asarray(self.f(x, up, p)) - np.asarray(self.f(x, um, p))) / (2 * du)
        return A, B

    def _glue(self, Jstate, Jmv, Jcon):      # Constraints.hpp:455-482
        return np.hstack([Jstate, Jmv @ self.Iz2u, Jcon.reshape(-1, 1)])

    # -- Constraints::getStateEqConstraints (Constraints.hpp:490-628) ----------------------------
    def state_eq(self, z, want_jac=True):
        nx, nu, ph = self.nx, self.nu, self.ph
        X, U, e = self.unwrap(z)
        c = np.zeros(ph * nx)
        Jx = np.zeros((ph * nx, ph * nx)); Jmv = np.zeros((ph * nx, ph * nu))
        Sx = np.diag(1.0 / self.state_scaling); Tx = np.diag(self.state_scaling); I = np.eye(nx)
        for i in range(ph):
            r = slice(i * nx, (i + 1) * nx)
            xk, uk, xk1 = X[i], U[i], X[i + 1]
            if self.continuous:
                h = self.Ts / 2.0
                c[r] = (xk + h * (np.asarray(self.f(xk, uk, i)) + np.asarray(self.f(xk1, uk, i))) - xk1) / self.state_scaling
                if want_jac:
                    Ak, Bk = self._state_jac(xk, uk, i)
                    Ak1, Bk1 = self._state_jac(xk1, uk, i)
                    if i > 0:
                        Jx[r, (i - 1) * nx:i * nx] = I + h * Sx @ Ak @ Tx
                    Jx[r, i * nx:(i + 1) * nx] = -I + h * Sx @ Ak1 @ Tx
                    Jmv[r, i * nu:(i + 1) * nu] = h * Sx @ (Bk + Bk1)
            else:
                c[r] = (xk1 - np.asarray(self.f(xk, uk, i))) / self.state_scaling
                if want_jac:
                    Ak, Bk = self._state_jac(xk, uk, i)
                    Jx[r, i * nx:(i + 1) * nx] = I
                    if i > 0:
                        Jx[r, (i - 1) * nx:i * nx] = -(Sx @ Ak @ Tx)
                    Jmv[r, i * nu:(i + 1) * nu] = -(Sx @ Bk)
        if not want_jac:
            return c, np.zeros((ph * nx, self.nz))
        return c, self._glue(Jx, Jmv, np.zeros(ph * nx))

    # -- Constraints::evaluateIneq + computeIneqJacobian (Constraints.hpp:211-316, 641-721) -------
    def user_ineq(self, z):
        nx, nu, ph = self.nx, self.nu, self.ph
        X, U, e = self.unwrap(z)
        fu = lambda X_, U_, e_: np.asarray(self.ineq_fun(X_, self.outputs(X_, U_), U_, e_), float)
        g0 = fu(X, U, e)
        Jx = np.zeros((self.ineq, ph * nx)); Jmv = np.zeros((self.ineq, ph * nu))
        Xa = np.maximum(np.abs(X), 1.0); Ua = np.maximum(np.abs(U), 1.0)
        lin = lambda M, j: M.reshape(-1, order="F")[j]
        Xp = X.copy(); Up = U.copy()
        for i in range(ph):
            for j in range(nx):
                dx = DV * lin(Xa, j)
                Xp[i + 1, j] += dx; fp = fu(Xp, Up, e)
                Xp[i + 1, j] -= 2 * dx; fm = fu(Xp, Up, e)
                Xp[i + 1, j] += dx
                Jx[:, i * nx + j] = (fp - fm) / (2 * dx)
        for i in range(ph):                      # every input row on its own, no pairing of the last one (:684-706)
            for j in range(nu):
                du = DV * lin(Ua, j)
                Up[i, j] += du; fp = fu(Xp, Up, e)
                Up[i, j] -= 2 * du; fm = fu(Xp, Up, e)
                Up[i, j] += du
                Jmv[:, i * nu + j] = (fp - fm) / (2 * du)
        de = max(DV, abs(e)) * DV
        Je = (fu(Xp, Up, e + de) - fu(Xp, Up, e - de)) / (2 * de)
        J = self._glue(Jx, Jmv, Je)
        J[:, :ph * nx] *= np.tile(self.state_scaling, ph)[None, :]      # :269-284
        return g0, J

    # -- Constraints::evaluateEq + computeEqJacobian (Constraints.hpp:365-442, 731-832) ----------
    def user_eq(self, z):
        """user equalities h(X, U) = 0 and their central-difference Jacobian.  Unlike the inequality Jacobian the step is
        taken from the perturbed element itself for the states (:751), from row ph-1 for every input step (:780, :806), and
        the last input row is perturbed together with its copy (:803-829)."""
        nx, nu, ph = self.nx, self.nu, self.ph
        X, U, e = self.unwrap(z)
        fu = lambda X_, U_: np.asarray(self.eq_fun(X_, U_), float)
        h0 = fu(X, U)
        Jx = np.zeros((self.eq, ph * nx)); Jmv = np.zeros((self.eq, ph * nu))
        Xa = np.maximum(np.abs(X), 1.0); Ua = np.maximum(np.abs(U), 1.0)
        Xp = X.copy(); Up = U.copy()
        for i in range(ph):
            for j in range(nx):
                dx = DV * Xa[i + 1, j]
                Xp[i + 1, j] += dx; fp = fu(Xp, Up)
                Xp[i + 1, j] -= 2 * dx; fm = fu(Xp, Up)
                Xp[i + 1, j] += dx
                Jx[:, i * nx + j] = (fp - fm) / (2 * dx)
        for i in range(ph):
            rows = [i] if i < ph - 1 else [ph - 1, ph]
            for j in range(nu):
                du = DV * Ua[ph - 1, j]
                for r in rows: Up[r, j] += du
                fp = fu(Xp, Up)
                for r in rows: Up[r, j] -= 2 * du
                fm = fu(Xp, Up)
                for r in rows: Up[r, j] += du
                Jmv[:, i * nu + j] = (fp - fm) / (2 * du)
        J = self._glue(Jx, Jmv, np.zeros(self.eq))
        J[:, :ph * nx] *= np.tile(self.state_scaling, ph)[None, :]      # :399-413
        return h0, J

    # -- NLOptimizer::run, cold start (NLOptimizer.hpp:412-638) with scipy's SLSQP ---------------
    def solve(self, x0, u0, max_iter=100, hard=True, lb_x=None, ub_x=None, lb_u=None, ub_u=None):
        from scipy.optimize import minimize
        nx, nu, ph, ch = self.nx, self.nu, self.ph, self.ch
        self.x0 = np.asarray(x0, float)
        # cold start (NLOptimizer.hpp:431-451): x0 and u0 replicated as they are -- no scaling applied to the guess
        z0 = np.concatenate([np.tile(self.x0, ph), np.tile(np.asarray(u0, float), ch), [0.0]])
        lo = np.full(self.nz, -np.inf); hi = np.full(self.nz, np.inf)
        if lb_x is not None:
            lo[:ph * nx] = np.tile(lb_x, ph); hi[:ph * nx] = np.tile(ub_x, ph)
        if lb_u is not None:
            lo[ph * nx:ph * nx + ch * nu] = np.tile(lb_u, ch); hi[ph * nx:ph * nx + ch * nu] = np.tile(ub_u, ch)
        if hard:
            lo[-1] = hi[-1] = 0.0                       # NLOptimizer.hpp:182-186
        cons = [{"type": "eq", "fun": lambda z: self.state_eq(z, False)[0], "jac": lambda z: self.state_eq(z, True)[1]}]
        if self.ineq_fun is not None:
            cons.append({"type": "ineq", "fun": lambda z: -self.user_ineq(z)[0], "jac": lambda z: -self.user_ineq(z)[1]})
        if self.eq_fun is not None:
            cons.append({"type": "eq", "fun": lambda z: self.user_eq(z)[0], "jac": lambda z: self.user_eq(z)[1]})
        r = minimize(lambda z: self.objective(z, False)[0], z0, jac=lambda z: self.objective(z, True)[1], method="SLSQP",
                     bounds=list(zip(lo, hi)), constraints=cons, options={"maxiter": max_iter, "ftol": 1e-12})
        X, U, e = self.unwrap(r.x)
        return dict(z=r.x, cmd=U[0].copy(), cost=float(r.fun), X=X, U=U, nit=int(r.nit), success=bool(r.success), message=str(r.message))


# ------------------------------------------------------------------------------------------------
# the reference's example models (data + formulas; examples/vanderpol_ex.cpp, examples/ugv_ex.cpp)
# ------------------------------------------------------------------------------------------------
def vanderpol(ph=10, ch=5, Ts=0.1):
    """examples/vanderpol_ex.cpp:9-65: 2 states, 1 input, cost sum x^2 + sum u^2, u_i <= 0.5"""
    m = NlmpcRef(2, 1, 2, ph, ch, ph + 1)
    m.continuous = True; m.Ts = Ts
    m.f = lambda x, u, p: np.array([(1.0 - x[1] * x[1]) * x[0] - x[1] + u[0], x[0]])
    m.cost = lambda X, Y, U, e: np.sum(X * X) + np.sum(U * U)
    m.ineq_fun = lambda X, Y, U, e: U[:, 0] - 0.5
    return m


def vanderpol_rate(ph=10, ch=5, Ts=0.1, rate=0.1):
    """the Van der Pol example with a rate limit |u_i - u_{i-1}| <= rate next to u_i <= 0.5 -- rows with two entries, several rows on one input
    (no reference example has them; mpcx::models::VanDerPolRate).  Rows grouped by step: [u_i - 0.5, u_i - u_{i-1} - rate, u_{i-1} - u_i - rate],
    step 0 comparing u_0 with itself"""
    m = vanderpol(ph, ch, Ts)
    m.ineq = 3 * (ph + 1)

    def ineq(X, Y, U, e):
        u = U[:, 0]
        du = u - np.concatenate([[u[0]], u[:-1]])
        return np.stack([u - 0.5, du - rate, -du - rate], axis=1).reshape(-1)
    m.ineq_fun = ineq
    return m


def vanderpol_terminal(ph=10, ch=5, Ts=0.1):
    """the Van der Pol example with a terminal equality x(ph) = 0 -- the textbook use of NLMPC::setEqConFunction
    (NLMPC.hpp:246-262); no reference example sets one"""
    m = vanderpol(ph, ch, Ts)
    m.eq = 2
    m.eq_fun = lambda X, U: X[ph].copy()
    return m


def ugv_matrices(Ts=0.1):
    """examples/ugv_ex.cpp:32-57: planar double integrator (px, py, vx, vy; ax, ay), zero-order hold"""
    A = np.eye(4); A[0, 2] = A[1, 3] = Ts
    B = np.zeros((4, 2)); B[0, 0] = B[1, 1] = 0.5 * Ts * Ts; B[2, 0] = B[3, 1] = Ts
    return A, B


def ugv(ph=30, ch=30, v_pref=(0.7071067811865476, 0.7071067811865476)):
    """examples/ugv_ex.cpp:12-137 at the horizon of SURVEY.md 8(d) config 3: discrete double integrator,
    two circular obstacles g = r - |p - p_obs| <= 0 at every step (:108-124), cost
    1e3 |v - v_pref|^2 + 1e-2 |u|^2 + 1e-5 e^2 (:86-104); v_pref is read uninitialised on the first
    solve in the example (:88 vs :159): here the unit vector towards yref = (2, 2)."""
    A, B = ugv_matrices()
    obs = np.array([[2.0, 1.0, 0.3], [1.0, 1.0, 0.3]])
    vp = np.asarray(v_pref, float)
    m = NlmpcRef(4, 2, 4, ph, ch, (ph + 1) * 2)
    m.continuous = False
    m.f = lambda x, u, p: A @ x + B @ u
    m.cost = lambda X, Y, U, e: 1e3 * np.sum((X[:, 2:4] - vp[None, :]) ** 2) + 1e-2 * np.sum(U * U) + 1e-5 * e * e

    def ineq(X, Y, U, e):
        g = np.zeros((X.shape[0], 2))
        for k in range(2):
            g[:, k] = obs[k, 2] - np.sqrt((X[:, 0] - obs[k, 0]) ** 2 + (X[:, 1] - obs[k, 1]) ** 2)
        return g.reshape(-1)
    m.ineq_fun = ineq
    return m


def oscillators(N=6, ph=20, ch=10, Ts=0.1, mu=1.0, k=0.1):
    """examples/networked_oscillators_ex.cpp:5-76: N Van der Pol oscillators with diffusive coupling k, continuous
    time, cost sum x^2 + sum u^2, u_ij <= 0.5 on every step"""
    m = NlmpcRef(2 * N, N, 2 * N, ph, ch, (ph + 1) * N)
    m.continuous = True; m.Ts = Ts

    def f(x, u, p):
        dx = np.zeros(2 * N)
        pos = x[0::2]
        for i in range(N):
            dx[2 * i] = x[2 * i + 1]
            a = mu * (1 - x[2 * i] * x[2 * i]) * x[2 * i + 1] - x[2 * i] + u[i]
            for j in range(N):
                if i != j:
                    a += k * (pos[j] - x[2 * i])
            dx[2 * i + 1] = a
        return dx
    m.f = f
    m.cost = lambda X, Y, U, e: np.sum(X * X) + np.sum(U * U)
    m.ineq_fun = lambda X, Y, U, e: (U - 0.5).reshape(-1)
    return m
