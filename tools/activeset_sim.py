"""CPU study aid (numpy, no GPU): rounds the working-set iteration of lmpc_solve needs on the config-2 batch under different
repair rules.  Uses the host set-up of a host-only handle (mpcx_lmpc_debug_get) for the condensed QP; the iteration itself is
restated here in numpy, per instance, exactly as the kernel runs it: solve on the working set through the
Schur complement Y[A,A], check the KKT conditions, repair.  Not product code, not the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libmpc_amd.workloads import quadrotor_lmpc, quadrotor_batch


def setup(ph, B, first=0):
    c = quadrotor_lmpc(ph, device=-1)
    d = c.debug_get("dims").astype(int)
    nz, mg, ldz, ldg, ldy = d[:5]
    dm = c.debug_get("dims_maps").astype(int)
    kin, nxp, nup, nyp, ione, nz16, mg16, ns, ns16, kq16, rowsA, ldy16 = dm
    MA = c.debug_get("MA1").reshape(kin, rowsA).T          # column-major rowsA x kin
    Y = c.debug_get("Y").reshape(ldy, ldy).T
    lw, uw = c.debug_get("lw")[:nz], c.debug_get("uw")[:nz]
    lg0, ug0 = c.debug_get("lg0")[:mg], c.debug_get("ug0")[:mg]
    x0, u0, yref = quadrotor_batch(B, first)
    vin = np.zeros((B, kin))
    vin[:, :12] = x0; vin[:, nxp:nxp + 4] = u0; vin[:, nxp + nup:nxp + nup + 12] = yref; vin[:, ione] = 1.0
    out = vin @ MA.T
    f = out[:, :nz]; goff = out[:, nz16:nz16 + mg]
    idx = np.r_[np.arange(nz), ldz + np.arange(mg)]
    Yc = Y[np.ix_(idx, idx)]                               # compact (nz+mg) x (nz+mg)
    t = -(f @ Yc[:nz, :])                                  # [t0 ; G t0] = -Y[:, :nz] f  (Y symmetric)
    lo = np.concatenate([np.broadcast_to(lw, (B, nz)), lg0[None, :] - goff], axis=1)
    hi = np.concatenate([np.broadcast_to(uw, (B, nz)), ug0[None, :] - goff], axis=1)
    return Yc, t, lo, hi


def solve_ws(Y, t, lo, hi, act):
    A = np.flatnonzero(act)
    if len(A) == 0:
        return t.copy(), np.zeros(0), A
    b = np.where(act[A] < 0, lo[A], hi[A])
    lam = np.linalg.solve(Y[np.ix_(A, A)], t[A] - b)
    return t - Y[:, A] @ lam, lam, A


def run(Y, t, lo, hi, rule, theta=0.3, maxr=40):
    """returns rounds; rule: 'cur' (drop-first then add >= theta max), 'pdas' (drop and add in one round), 'pdas_theta'"""
    ptol = 1e-8
    tol_lo = lo - ptol * np.maximum(1, np.abs(lo)); tol_hi = hi + ptol * np.maximum(1, np.abs(hi))
    act = np.where(t < tol_lo, -1, np.where(t > tol_hi, 1, 0))
    seen = set()
    for rd in range(1, maxr + 1):
        key = act.tobytes()
        if key in seen:
            return -rd          # cycle
        seen.add(key)
        w, lam, A = solve_ws(Y, t, lo, hi, act)
        dtol = 1e-9 * (np.abs(lam).max() if len(lam) else 0) + 1e-300
        wrong = np.zeros(len(A), bool)
        if len(A):
            wrong = ((act[A] < 0) & (lam > dtol)) | ((act[A] > 0) & (lam < -dtol))
        viol = np.maximum(np.maximum(tol_lo - w, w - tol_hi), 0.0)
        viol[A] = 0
        if rule == 'cur':
            if wrong.any():
                act[A[wrong]] = 0
                continue
            if viol.max() <= 0:
                return rd
            add = (viol > 0) & (viol >= theta * viol.max())
            act[add] = np.where(w[add] < lo[add], -1, 1)
        elif rule == 'pdas':
            if not wrong.any() and viol.max() <= 0:
                return rd
            act[A[wrong]] = 0
            add = (viol > 0) & (viol >= theta * viol.max())
            act[add] = np.where(w[add] < lo[add], -1, 1)
        elif rule == 'pdas_scaled':
            # violations scaled by 1/sqrt(Y_ii) (distance in the dual metric)
            if not wrong.any() and viol.max() <= 0:
                return rd
            act[A[wrong]] = 0
            sv = viol / np.sqrt(np.diag(Y))
            add = (sv > 0) & (sv >= theta * sv.max())
            act[add] = np.where(w[add] < lo[add], -1, 1)
    return -maxr


if __name__ == "__main__":
    ph = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    Y, T, LO, HI = setup(ph, B)
    for rule, theta in [('cur', 0.3), ('pdas', 0.3), ('pdas', 0.0), ('pdas', 0.1), ('pdas', 0.5), ('pdas_scaled', 0.3), ('pdas_scaled', 0.0)]:
        r = np.array([run(Y, T[i], LO[i], HI[i], rule, theta) for i in range(B)])
        ok = r > 0
        print("%-12s theta %.1f: mean %.2f  p90 %d  max %d  cycles %d  hist %s" % (
            rule, theta, r[ok].mean(), np.percentile(r[ok], 90), r[ok].max(), (~ok).sum(), np.bincount(r[ok])[:20]))


def run2(Y, t, lo, hi, theta_add=0.2, theta0=0.0, drop_frac=0.0, maxr=40):
    """simultaneous drop + add with separate thresholds; initial set = rows violated at t0 by >= theta0 * max"""
    ptol = 1e-8
    tol_lo = lo - ptol * np.maximum(1, np.abs(lo)); tol_hi = hi + ptol * np.maximum(1, np.abs(hi))
    v0 = np.maximum(np.maximum(tol_lo - t, t - tol_hi), 0.0)
    act = np.where((v0 > 0) & (v0 >= theta0 * v0.max()), np.where(t < lo, -1, 1), 0)
    seen = set()
    na_hist = []
    for rd in range(1, maxr + 1):
        key = act.tobytes()
        if key in seen:
            return -rd, na_hist
        seen.add(key)
        w, lam, A = solve_ws(Y, t, lo, hi, act)
        na_hist.append(len(A))
        dtol = 1e-9 * (np.abs(lam).max() if len(lam) else 0) + 1e-300
        bad = np.zeros(len(A))
        if len(A):
            bad = np.where(act[A] < 0, lam, -lam)          # > dtol: wrong sign
        wrong = bad > dtol
        if drop_frac > 0 and wrong.any():
            wrong = bad >= max(dtol, drop_frac * bad.max())
        viol = np.maximum(np.maximum(tol_lo - w, w - tol_hi), 0.0)
        viol[A] = 0
        if not (bad > dtol).any() and viol.max() <= 0:
            return rd, na_hist
        act[A[wrong]] = 0
        add = (viol > 0) & (viol >= theta_add * viol.max())
        act[add] = np.where(w[add] < lo[add], -1, 1)
    return -maxr, na_hist


def study2(ph, B):
    Y, T, LO, HI = setup(ph, B)
    for ta, t0, df in [(0.3, 0, 0), (0.2, 0, 0), (0.1, 0, 0), (0.2, 0.1, 0), (0.2, 0.3, 0), (0.2, 0, 0.1), (0.1, 0.05, 0)]:
        rr = [run2(Y, T[i], LO[i], HI[i], ta, t0, df) for i in range(B)]
        r = np.array([x[0] for x in rr]); ok = r > 0
        namax = max(max(x[1]) if x[1] else 0 for x in rr)
        print("add %.2f init %.2f drop %.2f: mean %.2f p90 %d max %d cycles %d maxna %d hist %s" % (
            ta, t0, df, r[ok].mean(), np.percentile(r[ok], 90), r[ok].max(), (~ok).sum(), namax, np.bincount(r[ok])[:16]))


def run3(Y, t, lo, hi, sched, theta0=0.3, maxr=40):
    """simultaneous drop + add, add threshold by round: sched(rd) -> theta"""
    ptol = 1e-8
    tol_lo = lo - ptol * np.maximum(1, np.abs(lo)); tol_hi = hi + ptol * np.maximum(1, np.abs(hi))
    v0 = np.maximum(np.maximum(tol_lo - t, t - tol_hi), 0.0)
    act = np.where((v0 > 0) & (v0 >= theta0 * v0.max()), np.where(t < lo, -1, 1), 0)
    seen = set()
    for rd in range(1, maxr + 1):
        key = act.tobytes()
        if key in seen:
            return -rd
        seen.add(key)
        w, lam, A = solve_ws(Y, t, lo, hi, act)
        dtol = 1e-9 * (np.abs(lam).max() if len(lam) else 0) + 1e-300
        bad = np.where(act[A] < 0, lam, -lam) if len(A) else np.zeros(0)
        wrong = bad > dtol
        viol = np.maximum(np.maximum(tol_lo - w, w - tol_hi), 0.0)
        viol[A] = 0
        if not wrong.any() and viol.max() <= 0:
            return rd
        act[A[wrong]] = 0
        add = (viol > 0) & (viol >= sched(rd) * viol.max())
        act[add] = np.where(w[add] < lo[add], -1, 1)
    return -maxr


def study3(ph, B):
    Y, T, LO, HI = setup(ph, B)
    scheds = {"0.2 flat": lambda r: 0.2, "0.2 then 0.05 from round 3": lambda r: 0.2 if r < 3 else 0.05, "0.2 then 0.1 from 3": lambda r: 0.2 if r < 3 else 0.1,
              "0.3,0.2,0.1,0.05..": lambda r: max(0.05, 0.4 - 0.1 * r), "0.1 flat": lambda r: 0.1, "0.2 then 0.02 from 4": lambda r: 0.2 if r < 4 else 0.02,
              "0.2 then 0 from 4": lambda r: 0.2 if r < 4 else 0.0}
    for name, f in scheds.items():
        for t0 in (0.3, 0.15):
            r = np.array([run3(Y, T[i], LO[i], HI[i], f, t0) for i in range(B)])
            ok = r > 0
            print("%-28s init %.2f: mean %.3f p99 %d max %d cycles %d hist %s" % (name, t0, r[ok].mean(), np.percentile(r[ok], 99), r[ok].max(), (~ok).sum(), np.bincount(r[ok])[1:12]))
