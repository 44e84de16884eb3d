"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference LMPC QP.

This module restates, in plain numpy, what libmpc++'s linear-MPC problem
builder encodes (reference: include/mpc/LMPC/ProblemBuilder.hpp) and how
LOptimizer unpacks a solution (include/mpc/LMPC/LOptimizer.hpp:305-347).
It is a *checker*: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.  The product path (libmpc_amd) never does.

Every function cites the reference lines it follows.  Nothing here is copied
from the reference; the formulation is re-derived from SURVEY.md Appendix A
and checked against the reference's own known answers
(test/LMPC/test_common.cpp:230-236, test/LMPC/test_constraints.cpp:183-294).

Row numbering of A/l/u (ProblemBuilder.hpp:814-822) -- the numbering in which
active-set indices are reported:
    [0, neq)                      dynamics equalities, neq = (ph+1)*na
    [neq, neq+nbox)               box on [x; x_u],     nbox = (ph+1)*na
    [.., +(ph+1)*ny)              output box
    [.., +ph*nu)                  delta-u box
    [.., +(ph+1))                 scalar constraint rows
"""
from __future__ import annotations

import numpy as np

INF = np.inf


class LmpcDims:
    def __init__(self, nx, nu, ndu, ny, ph, ch):
        self.nx, self.nu, self.ndu, self.ny, self.ph, self.ch = nx, nu, ndu, ny, ph, ch
        self.na = nx + nu
        self.nvar = (ph + 1) * self.na + ph * nu          # ProblemBuilder.hpp:70
        self.neq = (ph + 1) * self.na
        self.nineq = (ph + 1) * self.na + (ph + 1) * ny + ph * nu + (ph + 1)
        self.ncon = self.neq + self.nineq                 # ProblemBuilder.hpp:74
        # offsets inside the inequality block
        self.off_box = 0
        self.off_y = (ph + 1) * self.na
        self.off_du = self.off_y + (ph + 1) * ny
        self.off_s = self.off_du + ph * nu


class ProblemBuilderRef:
    """State and setters of the reference builder (ProblemBuilder.hpp:88-504)."""

    def __init__(self, nx, nu, ndu, ny, ph, ch):
        d = self.d = LmpcDims(nx, nu, ndu, ny, ph, ch)
        na = d.na
        self.ssA = np.zeros((na, na))
        self.ssB = np.zeros((na, nu))
        self.ssC = np.zeros((ny + nu, na))
        self.ssBv = np.zeros((na, ndu))
        self.ssDv = np.zeros((ny + nu, ndu))
        self.wOutput = np.zeros((ny, ph + 1))
        self.wU = np.zeros((nu, ph + 1))
        self.wDeltaU = np.zeros((nu, ph))
        self.minX = np.full((nx, ph + 1), -INF)
        self.maxX = np.full((nx, ph + 1), INF)
        self.minY = np.full((ny, ph + 1), -INF)
        self.maxY = np.full((ny, ph + 1), INF)
        self.minU = np.full((nu, ph), -INF)
        self.maxU = np.full((nu, ph), INF)
        self.sMin = np.full(ph + 1, -INF)
        self.sMax = np.full(ph + 1, INF)
        self.sX = np.zeros(nx)
        self.sU = np.zeros(nu)

    # -- model (ProblemBuilder.hpp:184-236) ---------------------------------
    def set_state_model(self, A, B, C):
        d = self.d
        nx, nu, ny = d.nx, d.nu, d.ny
        self.ssA[:] = 0
        self.ssA[:nx, :nx] = A
        self.ssA[:nx, nx:] = B
        self.ssA[nx:, nx:] = np.eye(nu)
        self.ssB[:nx, :] = B
        self.ssB[nx:, :] = np.eye(nu)
        self.ssC[:] = 0
        self.ssC[:ny, :nx] = C
        self.ssC[ny:, nx:] = np.eye(nu)

    def set_exogenous(self, Bd, Dd):
        d = self.d
        self.ssBv[:] = 0
        self.ssBv[:d.nx, :] = Bd
        self.ssDv[:] = 0
        self.ssDv[:d.ny, :] = Dd

    # -- weights (ProblemBuilder.hpp:247-297): user column k -> internal k+1,
    #    internal column 0 := user column 0 --------------------------------
    def set_objective_mat(self, OW, UW, DUW):
        self.wOutput[:, 1:] = OW
        self.wOutput[:, 0] = OW[:, 0]
        self.wU[:, 1:] = UW
        self.wU[:, 0] = UW[:, 0]
        self.wDeltaU[:] = DUW

    def set_objective_idx(self, index, ow, uw, duw):
        self.wOutput[:, index + 1] = ow
        self.wU[:, index + 1] = uw
        if index == 0:
            self.wOutput[:, 0] = ow
            self.wU[:, 0] = uw
        self.wDeltaU[:, index] = duw

    # -- bounds (ProblemBuilder.hpp:378-504) --------------------------------
    def set_state_bounds_mat(self, lo, hi):
        self.minX[:, 1:] = lo
        self.minX[:, 0] = lo[:, 0]
        self.maxX[:, 1:] = hi
        self.maxX[:, 0] = hi[:, 0]

    def set_state_bounds_idx(self, index, lo, hi):
        self.minX[:, index + 1] = lo
        self.maxX[:, index + 1] = hi
        if index == 0:
            self.minX[:, 0] = lo
            self.maxX[:, 0] = hi

    def set_output_bounds_mat(self, lo, hi):
        self.minY[:, 1:] = lo
        self.minY[:, 0] = lo[:, 0]
        self.maxY[:, 1:] = hi
        self.maxY[:, 0] = hi[:, 0]

    def set_output_bounds_idx(self, index, lo, hi):
        self.minY[:, index + 1] = lo
        self.maxY[:, index + 1] = hi
        if index == 0:
            self.minY[:, 0] = lo
            self.maxY[:, 0] = hi

    def set_input_bounds_mat(self, lo, hi):
        # nu x ch in, replicated past the control horizon (ProblemBuilder.hpp:402-410)
        ch, ph = self.d.ch, self.d.ph
        self.minU[:, :ch] = lo
        self.maxU[:, :ch] = hi
        if ch < ph:
            self.minU[:, ch:] = lo[:, ch - 1:ch]
            self.maxU[:, ch:] = hi[:, ch - 1:ch]

    def set_input_bounds_idx(self, index, lo, hi):
        self.minU[:, index] = lo
        self.maxU[:, index] = hi

    # -- scalar constraint (ProblemBuilder.hpp:310-365) ---------------------
    def set_scalar_vec(self, smin, smax, X, U):
        self.sMin[1:] = smin
        self.sMin[0] = smin[0]
        self.sMax[1:] = smax
        self.sMax[0] = smax[0]
        self.sX[:] = X
        self.sU[:] = U

    def set_scalar_idx(self, index, smin, smax, X, U):
        self.sMin[index + 1] = smin
        self.sMax[index + 1] = smax
        if index == 0:
            self.sMin[0] = smin
            self.sMax[0] = smax
        self.sX[:] = X       # the multiplier row is rewritten for every step
        self.sU[:] = U

    # -- time-invariant part (ProblemBuilder.hpp:642-825) -------------------
    def build(self):
        d = self.d
        nx, nu, ny, ph, ch, na = d.nx, d.nu, d.ny, d.ph, d.ch, d.na
        P = np.zeros((d.nvar, d.nvar))
        for i in range(ph + 1):
            W = np.diag(np.concatenate([self.wOutput[:, i], self.wU[:, i]]))
            P[i * na:(i + 1) * na, i * na:(i + 1) * na] = self.ssC.T @ W @ self.ssC
            if i < ph:
                o = (ph + 1) * na + i * nu
                P[o:o + nu, o:o + nu] = np.diag(self.wDeltaU[:, i])

        A = np.zeros((d.ncon, d.nvar))
        # dynamics: row-block 0: -xi_0 ; row-block i: ssA xi_{i-1} - xi_i + ssB du_{i-1}
        for i in range(ph + 1):
            A[i * na:(i + 1) * na, i * na:(i + 1) * na] = -np.eye(na)
            if i > 0:
                A[i * na:(i + 1) * na, (i - 1) * na:i * na] += self.ssA
                o = (ph + 1) * na + (i - 1) * nu
                A[i * na:(i + 1) * na, o:o + nu] = self.ssB
        r0 = d.neq
        # box rows on [x; x_u]
        A[r0:r0 + (ph + 1) * na, :(ph + 1) * na] = np.eye((ph + 1) * na)
        # output rows
        for i in range(ph + 1):
            A[r0 + d.off_y + i * ny: r0 + d.off_y + (i + 1) * ny, i * na:(i + 1) * na] = self.ssC[:ny, :]
        # delta-u rows
        A[r0 + d.off_du: r0 + d.off_du + ph * nu, (ph + 1) * na:] = np.eye(ph * nu)
        # scalar rows
        for i in range(ph + 1):
            A[r0 + d.off_s + i, i * na:(i + 1) * na] = np.concatenate([self.sX, self.sU])

        lineq = np.zeros(d.nineq)
        uineq = np.zeros(d.nineq)
        for i in range(ph + 1):
            k = min(i, ph - 1)      # x_u(i)=u(i-1) takes input-bound column min(i,ph-1) (:735-749)
            lineq[i * na:(i + 1) * na] = np.concatenate([self.minX[:, i], self.minU[:, k]])
            uineq[i * na:(i + 1) * na] = np.concatenate([self.maxX[:, i], self.maxU[:, k]])
        lineq[d.off_y:d.off_du] = self.minY.T.reshape(-1)     # column-major flatten (:755-765)
        uineq[d.off_y:d.off_du] = self.maxY.T.reshape(-1)
        for i in range(ph):
            pinned = i > ch                                    # strict (:782-793)
            lineq[d.off_du + i * nu: d.off_du + (i + 1) * nu] = 0.0 if pinned else -INF
            uineq[d.off_du + i * nu: d.off_du + (i + 1) * nu] = 0.0 if pinned else INF
        lineq[d.off_s:] = self.sMin
        uineq[d.off_s:] = self.sMax
        self.P, self.A, self.lineq, self.uineq = P, A, lineq, uineq
        return P, A

    # -- per-solve vectors (ProblemBuilder.hpp:528-633) ---------------------
    def get(self, x0, u0, yRef, uRef, duRef, dMeas):
        d = self.d
        nx, nu, ny, ph, na = d.nx, d.nu, d.ny, d.ph, d.na
        q = np.zeros(d.nvar)
        leq = np.zeros(d.neq)
        off = np.zeros(d.nineq)
        for i in range(ph + 1):
            k = max(i - 1, 0)
            dk = dMeas[:, k] if d.ndu > 0 else np.zeros(0)
            eref = np.concatenate([yRef[:, k], uRef[:, k]])
            W = np.concatenate([self.wOutput[:, i], self.wU[:, i]])
            q[i * na:(i + 1) * na] = self.ssC.T @ (W * (-eref + self.ssDv @ dk))
            if i < ph:
                o = (ph + 1) * na + i * nu
                q[o:o + nu] = -(self.wDeltaU[:, i] * duRef[:, k])
            if i > 0:
                leq[i * na:(i + 1) * na] = -(self.ssBv @ dk)
            off[d.off_y + i * ny: d.off_y + (i + 1) * ny] = -(self.ssDv[:ny, :] @ dk)
        leq[:nx] = -x0
        leq[nx:na] = -u0
        with np.errstate(invalid="ignore"):
            l = np.concatenate([leq, self.lineq + off])
            u = np.concatenate([leq, self.uineq + off])
        return q, l, u


def unpack_solution(d: LmpcDims, z, C, Dd, dMeas):
    """LOptimizer.hpp:305-347: sequences and cmd from the QP primal vector."""
    nx, nu, ny, ph, na = d.nx, d.nu, d.ny, d.ph, d.na
    state = np.zeros((ph + 1, nx))
    inp = np.zeros((ph + 1, nu))
    out = np.zeros((ph + 1, ny))
    for i in range(ph + 1):
        state[i] = z[i * na:i * na + nx]
        j = i + 1 if i + 1 < ph + 1 else i
        inp[i] = z[j * na + nx:(j + 1) * na]
        k = max(i - 1, 0)
        dk = dMeas[:, k] if d.ndu > 0 else np.zeros(0)
        out[i] = C @ state[i] + (Dd @ dk if d.ndu > 0 else 0.0)
    return state, out, inp


# ---------------------------------------------------------------------------
# Workloads
# ---------------------------------------------------------------------------

def quadrotor_model():
    """Model of examples/quadrotor_ex.cpp:19-48 (values transcribed as data)."""
    Ad = np.eye(12)
    Ad[0, 6] = Ad[1, 7] = Ad[2, 8] = 0.1
    Ad[3, 0] = 0.0488; Ad[3, 6] = 0.0016; Ad[3, 9] = 0.0992
    Ad[4, 1] = -0.0488; Ad[4, 7] = -0.0016; Ad[4, 10] = 0.0992
    Ad[5, 11] = 0.0992
    Ad[9, 0] = 0.9734; Ad[9, 6] = 0.0488; Ad[9, 9] = 0.9846
    Ad[10, 1] = -0.9734; Ad[10, 7] = -0.0488; Ad[10, 10] = 0.9846
    Ad[11, 11] = 0.9846
    Bd = np.array([
        [0, -0.0726, 0, 0.0726],
        [-0.0726, 0, 0.0726, 0],
        [-0.0152, 0.0152, -0.0152, 0.0152],
        [0, -0.0006, -0.0000, 0.0006],
        [0.0006, 0, -0.0006, 0],
        [0.0106, 0.0106, 0.0106, 0.0106],
        [0, -1.4512, 0, 1.4512],
        [-1.4512, 0, 1.4512, 0],
        [-0.3049, 0.3049, -0.3049, 0.3049],
        [0, -0.0236, 0, 0.0236],
        [0.0236, 0, -0.0236, 0],
        [0.2107, 0.2107, 0.2107, 0.2107]], dtype=float)
    Cd = np.eye(12)
    return Ad, Bd, Cd


def quadrotor_builder(ph, ch=None):
    """Controller set-up of examples/quadrotor_ex.cpp:52-93 at horizon ph."""
    ch = ph if ch is None else ch
    nx, nu, ndu, ny = 12, 4, 4, 12
    b = ProblemBuilderRef(nx, nu, ndu, ny, ph, ch)
    Ad, Bd, Cd = quadrotor_model()
    b.set_state_model(Ad, Bd, Cd)
    ow = np.array([0, 0, 10, 10, 10, 10, 0, 0, 0, 5, 5, 5], dtype=float)
    uw = np.full(4, 0.1)
    duw = np.zeros(4)
    xmin = np.full(12, -INF); xmax = np.full(12, INF)
    xmin[0] = xmin[1] = -np.pi / 6; xmax[0] = xmax[1] = np.pi / 6
    xmin[5] = -1.0
    umin = np.full(4, 9.6 - 10.5916); umax = np.full(4, 13 - 10.5916)
    for i in range(ph):             # slices {0,ph} go through the per-index setters
        b.set_objective_idx(i, ow, uw, duw)
        b.set_state_bounds_idx(i, xmin, xmax)
    for i in range(ch):
        b.set_input_bounds_idx(i, umin, umax)
    if ch < ph:
        # the example sets {0,Tch}; steps beyond keep the builder default (+-inf)
        pass
    b.build()
    return b


def splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    z = z ^ (z >> 31)
    return state, z


def quadrotor_batch(B, first=0):
    """Synthetic per-instance inputs of SURVEY.md section 8(d), config 2.

    SplitMix64 stream seeded 0x6d70632b2b + instance index; doubles are
    (r >> 11) * 2^-53.  Instance 0 is the reference test's exact input.
    Returns x0[B,12], u0[B,4], yref[B,12] (constant along the horizon).
    """
    x0 = np.zeros((B, 12)); u0 = np.zeros((B, 4)); yref = np.zeros((B, 12))
    for b in range(B):
        idx = first + b
        s = (0x6d70632b2b + idx) & 0xFFFFFFFFFFFFFFFF
        def uni(lo, hi):
            nonlocal s
            s, r = splitmix64(s)
            return lo + (hi - lo) * ((r >> 11) * 2.0 ** -53)
        for j in range(12):
            if j < 2:
                x0[b, j] = uni(-0.2, 0.2)
            elif j < 6:
                x0[b, j] = uni(-0.5, 0.5)
            else:
                x0[b, j] = uni(-0.3, 0.3)
        for j in range(4):
            u0[b, j] = uni(-0.5, 0.5)
        yref[b, 2] = uni(0.5, 1.5)
        if idx == 0:
            x0[b] = 0; u0[b] = 0; yref[b] = 0; yref[b, 2] = 1.0
    return x0, u0, yref
