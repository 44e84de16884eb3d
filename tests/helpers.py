"""Shared test helpers: drive the CPU oracle and the product with the same reference-style calls."""
import math

import numpy as np

from oracle.lmpc_oracle import OracleLMPC, default_params

INF = float("inf")


class OracleFrontEnd:
    """The reference's LMPC front-end logic (include/mpc/LMPC.hpp) on top of the C oracle:
    {-1,-1} slices go through the matrix setters, other slices through the per-index setters,
    exactly as LMPC.hpp:153-292,436-481,596-660 do.  Lets a test issue the same calls to the
    oracle and to libmpc_amd.LMPC."""

    def __init__(self, nx, nu, ndu, ny, ph, ch):
        self.nx, self.nu, self.ndu, self.ny, self.ph, self.ch = nx, nu, ndu, ny, ph, ch
        self.o = OracleLMPC(nx, nu, ndu, ny, ph, ch)
        self.yRef = np.zeros((ny, ph)); self.uRef = np.zeros((nu, ph)); self.duRef = np.zeros((nu, ph))
        self.dMeas = np.zeros((ndu, ph))

    @staticmethod
    def _unset(s):
        return s is None or tuple(s) == (-1, -1)

    def _pred_ok(self, s):
        a, b = s
        return not (a >= b or a > self.ph or b > self.ph or a < 0)

    def _ctrl_ok(self, s):
        a, b = s
        return not (a >= b or a > self.ch or b > self.ch or a < 0)

    def setStateSpaceModel(self, A, B, C):
        return bool(self.o.set_model(A, B, C))

    def setDisturbances(self, Bd, Dd):
        return bool(self.o.set_exogenous(np.asarray(Bd, float).reshape(self.nx, self.ndu),
                                         np.asarray(Dd, float).reshape(self.ny, self.ndu)))

    def setOptimizerParameters(self, **kw):
        self.o.params = default_params(**kw)

    def setObjectiveWeights(self, ow, uw, duw, slice=None):
        ow = np.asarray(ow, float)
        if ow.ndim == 2 and slice is None:
            return bool(self.o.set_objective(ow, uw, duw))
        if self._unset(slice):
            rep = lambda v: np.tile(np.asarray(v, float).reshape(-1, 1), (1, self.ph))
            return bool(self.o.set_objective(rep(ow), rep(uw), rep(duw)))
        if not self._pred_ok(slice):
            return False
        for i in range(slice[0], slice[1]):
            self.o.set_objective_idx(i, ow, uw, duw)
        return True

    def _bounds(self, lo, hi, cols, fmat, fidx, ok, slice):
        lo = np.asarray(lo, float)
        if lo.ndim == 2 and slice is None:
            return bool(fmat(lo, hi))
        if self._unset(slice):
            rep = lambda v: np.tile(np.asarray(v, float).reshape(-1, 1), (1, cols))
            return bool(fmat(rep(lo), rep(hi)))
        if not ok(slice):
            return False
        for i in range(slice[0], slice[1]):
            fidx(i, lo, hi)
        return True

    def setStateBounds(self, lo, hi, slice=None):
        return self._bounds(lo, hi, self.ph, self.o.set_state_bounds, self.o.set_state_bounds_idx, self._pred_ok, slice)

    def setInputBounds(self, lo, hi, slice=None):
        return self._bounds(lo, hi, self.ch, self.o.set_input_bounds, self.o.set_input_bounds_idx, self._ctrl_ok, slice)

    def setOutputBounds(self, lo, hi, slice=None):
        return self._bounds(lo, hi, self.ph, self.o.set_output_bounds, self.o.set_output_bounds_idx, self._pred_ok, slice)

    def setScalarConstraint(self, *args):
        if isinstance(args[4], (tuple, list)) or args[4] is None:
            smin, smax, X, U, sl = args
            if self._unset(sl):
                return bool(self.o.set_scalar(np.full(self.ph, smin), np.full(self.ph, smax), X, U))
            if not self._pred_ok(sl):
                return False
            for i in range(sl[0], sl[1]):
                self.o.set_scalar_idx(i, smin, smax, X, U)
            return True
        index, smin, smax, X, U = args
        if index >= self.ph:
            return False
        self.o.set_scalar_idx(index, smin, smax, X, U)
        return True

    def setReferences(self, y, u, du, slice=None):
        y = np.asarray(y, float)
        if y.ndim == 2 and slice is None:
            self.yRef[:] = y; self.uRef[:] = u; self.duRef[:] = du
            return True
        s = (0, self.ph) if self._unset(slice) else slice
        if not self._pred_ok(s):
            return False
        for i in range(s[0], s[1]):
            self.yRef[:, i] = y; self.uRef[:, i] = u; self.duRef[:, i] = du
        return True

    def setExogenousInputs(self, d, slice=None):
        d = np.asarray(d, float)
        if d.ndim == 2 and slice is None:
            self.dMeas[:] = d
            return True
        s = (0, self.ph) if self._unset(slice) else slice
        if not self._unset(slice) and not self._ctrl_ok(s):       # LMPC.hpp:571: control-horizon validity
            return False
        for i in range(s[0], s[1]):
            self.dMeas[:, i] = d
        return True

    def optimize(self, x0, u0, yRef=None, uRef=None, duRef=None, dMeas=None):
        """one cold-start LOptimizer::run; optional per-solve overrides of the references"""
        yR = self.yRef if yRef is None else yRef
        uR = self.uRef if uRef is None else uRef
        dR = self.duRef if duRef is None else duRef
        dM = self.dMeas if dMeas is None else dMeas
        return self.o.solve(x0, u0, yR, uR, dR, dM)


def configure_quadrotor(c, ph, ch=None):
    """examples/quadrotor_ex.cpp:52-93 through reference-style calls (works on both front-ends)."""
    from oracle.lmpc_numpy import quadrotor_model
    ch = ph if ch is None else ch
    Ad, Bd, Cd = quadrotor_model()
    assert c.setStateSpaceModel(Ad, Bd, Cd)
    assert c.setObjectiveWeights([0, 0, 10, 10, 10, 10, 0, 0, 0, 5, 5, 5], [0.1] * 4, [0] * 4, (0, ph))
    xmin = [-math.pi / 6, -math.pi / 6, -INF, -INF, -INF, -1] + [-INF] * 6
    xmax = [math.pi / 6, math.pi / 6] + [INF] * 10
    assert c.setStateBounds(xmin, xmax, (0, ph))
    assert c.setOutputBounds([-INF] * 12, [INF] * 12, (0, ph))
    assert c.setInputBounds([9.6 - 10.5916] * 4, [13 - 10.5916] * 4, (0, ch))
    yref = np.zeros(12); yref[2] = 1.0
    assert c.setReferences(yref, np.zeros(4), np.zeros(4), (0, ph))
    return c


def quadrotor_oracle(ph, ch=None, maximum_iteration=250):
    """the quadrotor controller on the raw C oracle (for batch drivers)"""
    f = OracleFrontEnd(12, 4, 4, 12, ph, ph if ch is None else ch)
    configure_quadrotor(f, ph, ch)
    f.o.params = default_params(maximum_iteration=maximum_iteration)
    return f.o


def random_lmpc_spec(seed, nx=3, nu=2, ndu=1, ny=2, ph=6, ch=3):
    """A small controller exercising every LMPC feature: disturbances, per-step weights, state /
    input / output bounds on slices, a scalar constraint, move blocking (ch < ph), references."""
    r = np.random.default_rng(seed)
    A = r.normal(size=(nx, nx)); A *= 0.9 / max(abs(np.linalg.eigvals(A)))
    spec = dict(dims=(nx, nu, ndu, ny, ph, ch), A=A, B=r.normal(size=(nx, nu)), C=r.normal(size=(ny, nx)),
                Bd=0.3 * r.normal(size=(nx, ndu)), Dd=0.2 * r.normal(size=(ny, ndu)),
                OW=r.uniform(0.5, 2.0, size=(ny, ph)), UW=r.uniform(0.05, 0.2, size=(nu, ph)),
                DUW=r.uniform(0.0, 0.1, size=(nu, ph)),
                umin=-0.6 * np.ones(nu), umax=0.7 * np.ones(nu),
                xmin=np.array([-2.0] + [-INF] * (nx - 1)), xmax=np.array([2.0] + [INF] * (nx - 1)),
                ymin=np.full(ny, -3.0), ymax=np.full(ny, 3.0),
                sX=r.normal(size=nx), sU=r.normal(size=nu), smin=-4.0, smax=4.0,
                yref=r.normal(size=(ny, ph)), uref=0.05 * r.normal(size=(nu, ph)), duref=0.01 * r.normal(size=(nu, ph)),
                dmeas=0.2 * r.normal(size=(ndu, ph)))
    return spec


def configure_random(c, spec):
    nx, nu, ndu, ny, ph, ch = spec["dims"]
    assert c.setStateSpaceModel(spec["A"], spec["B"], spec["C"])
    assert c.setDisturbances(spec["Bd"], spec["Dd"])
    assert c.setObjectiveWeights(spec["OW"], spec["UW"], spec["DUW"])
    assert c.setInputBounds(spec["umin"], spec["umax"], None)
    assert c.setStateBounds(spec["xmin"], spec["xmax"], (1, ph))
    assert c.setOutputBounds(spec["ymin"], spec["ymax"], (0, ph - 1))
    assert c.setScalarConstraint(spec["smin"], spec["smax"], spec["sX"], spec["sU"], (1, ph))
    assert c.setReferences(spec["yref"], spec["uref"], spec["duref"])
    assert c.setExogenousInputs(spec["dmeas"])
    return c


def bits_to_rows(words, m):
    """[B, W] int32 bitmap words -> list of sorted row-index arrays"""
    w = np.ascontiguousarray(np.asarray(words)).astype(np.uint32)
    out = []
    for b in range(w.shape[0]):
        bits = np.unpackbits(w[b].view(np.uint8), bitorder="little")[:m]
        out.append(np.nonzero(bits)[0])
    return out


def rel_err(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


def oracle_batch_parallel(ph, x0, u0, yref, want_active=False, workers=None, **kw):
    """solve_batch_constref of the C oracle over a thread pool (ctypes releases the GIL during the call; one oracle handle per
    thread): the whole benchmark batch in seconds instead of a sample of it"""
    import os
    from concurrent.futures import ThreadPoolExecutor
    B = len(x0)
    workers = workers or max(1, min(16, (os.cpu_count() or 1)))
    bounds = np.linspace(0, B, workers + 1).astype(int)
    def job(k):
        lo, hi = bounds[k], bounds[k + 1]
        if hi <= lo:
            return None
        return quadrotor_oracle(ph, **kw).solve_batch_constref(x0[lo:hi], u0[lo:hi], yref[lo:hi], want_active=want_active)
    with ThreadPoolExecutor(workers) as ex:
        parts = [r for r in ex.map(job, range(workers)) if r is not None]
    out = {}
    for k in parts[0]:
        if parts[0][k] is None or np.isscalar(parts[0][k]):
            out[k] = parts[0][k]
        else:
            out[k] = np.concatenate([p_[k] for p_ in parts])
    return out


def assert_matches_oracle(r, ref, o_neq, o_ncon, rtol_cmd=1e-5, rtol_cost=1e-7, check_active=True):
    """GPU batch result against oracle results: cmd, cost on the instances the oracle polished, active sets bit for bit there,
    loose on the others (the reference returns an eps = 1e-4 ADMM iterate there, the GPU the exact optimum); returns the number
    of unpolished instances"""
    cmd = r.cmd.cpu().numpy(); cost = r.cost.cpu().numpy(); st = r.status.cpu().numpy()
    pol = ref["polished"] == 1
    assert np.array_equal(st[ref["status"] == 0], np.zeros((ref["status"] == 0).sum(), dtype=st.dtype))
    scale = np.maximum(np.abs(ref["cmd"]).max(axis=1), 1e-12)
    err = np.abs(cmd - ref["cmd"]).max(axis=1) / scale
    assert err[pol].max() <= rtol_cmd, (err[pol].max(), int(np.argmax(err * pol)))
    if (~pol).any():
        assert err[~pol].max() <= 5e-2
    cerr = np.abs(cost - ref["cost"]) / np.maximum(1.0, np.abs(ref["cost"]))
    assert cerr[pol].max() <= rtol_cost, cerr[pol].max()
    if check_active:
        wl = np.ascontiguousarray(r.active_lower.cpu().numpy()).astype(np.uint32); wu = np.ascontiguousarray(r.active_upper.cpu().numpy()).astype(np.uint32)
        bl = np.unpackbits(wl.view(np.uint8).reshape(len(wl), -1), axis=1, bitorder="little")[:, :o_ncon]
        bu = np.unpackbits(wu.view(np.uint8).reshape(len(wu), -1), axis=1, bitorder="little")[:, :o_ncon]
        rl = (ref["active_lower"] != 0).astype(np.uint8); ru = (ref["active_upper"] != 0).astype(np.uint8)
        same = (bl[:, o_neq:] == rl[:, o_neq:]).all(axis=1) & (bu[:, o_neq:] == ru[:, o_neq:]).all(axis=1)
        assert same[pol].all(), int(np.argmin(same | ~pol))
    return int((~pol).sum())
