"""-m gpu: the oscillator controllers with TIGHT INPUT BOUNDS (NLMPC::setInputBounds, NLMPC.hpp:300-316 -> NLOptimizer.hpp:346-404) -- working
sets that fill the sub-problem's variables (58 .. 60 of 60 inputs on a bound at six oscillators, 100 .. 120 of 120 at eight).  Round 5's last
kernel commit sent the six-oscillator shape to the inverse form of the working set's Schur complement and lost 2 .. 5 % of such instances
(solver status -1 / -3 / -4 on feasible problems); no test had an oscillator's bounds in the working set.  These do, in every form of the
kernel and with the inverse form forced both ways:

  * the committed oracle solutions (tests/golden/nlmpc_oracle_solutions.json, keys *_bounds; generator make_nlmpc_bounds_golden.py): status,
    u* (1e-5 of max(1, |u*|)), cost (1e-8), inputs inside the bounds;
  * batches of 256 .. 1024: EVERY instance solved, feasible, inside the bounds, and (six oscillators: every instance; eight: a sample) equal to
    the oracle's SLSQP run here on the host's cores; the restated problem's KKT conditions at the returned point on a sample;
  * how many instances left the inverse form on the way (a sub-problem failed the check of its working rows) is counted and printed: the
    path the round-5 regression went through is exercised, and its results are those of a handle planned without the inverse form."""
import json
import os

import numpy as np
import pytest

from oracle import nlmpc_c
from oracle import nlmpc_numpy as ref

pytestmark = pytest.mark.gpu

SHAPES = dict(osc6=dict(N=6, ph=20, ch=10, key="oscillators6_ph20_ch10_bounds"), osc8=dict(N=8, ph=30, ch=15, key="oscillators8_ph30_ch15_bounds"))
ENVS = ("MPCX_NLMPC_FORM", "MPCX_NLMPC_WAVES", "MPCX_NLMPC_BLOCKS", "MPCX_NLMPC_MINV", "MPCX_NLMPC_CARRY")
# The optimal cost is compared at 2e-7 relative here (1e-8 in the tests without active bounds): the solve stops on a step of 1e-6 (tol_step), and along
# an input that sits on its bound the cost is LINEAR in the distance -- multiplier (up to ~10 here) x 1e-6 / cost (~100) = 1e-7; measured: 2e-10 typical,
# 7.5e-8 the worst of ~4000 comparisons (an input that ends 1e-7 inside its bound).  u* itself is compared at north_star's 1e-5.
COST_RTOL = 2e-7
# variants: the switches are read when a handle is created
VARIANTS = {"default": {}, "wg-factor": dict(MPCX_NLMPC_FORM="wg", MPCX_NLMPC_MINV="0"), "wg-inverse": dict(MPCX_NLMPC_FORM="wg", MPCX_NLMPC_MINV="1"),
            "wg-inverse-not-carried": dict(MPCX_NLMPC_FORM="wg", MPCX_NLMPC_MINV="1", MPCX_NLMPC_CARRY="0"), "wave": dict(MPCX_NLMPC_FORM="wave")}


def _controller(monkeypatch, shape, ub, variant, iters=200):
    from libmpc_amd.nlmpc import NLMPC, NLParameters, OSCILLATORS6, OSCILLATORS8
    for k in ENVS:
        monkeypatch.delenv(k, raising=False)
    for k, v in VARIANTS[variant].items():
        monkeypatch.setenv(k, v)
    s = SHAPES[shape]
    c = NLMPC(OSCILLATORS6 if s["N"] == 6 else OSCILLATORS8, s["ph"], s["ch"], 0.1)
    c.setOptimizerParameters(NLParameters(maximum_iteration=iters))
    if ub is not None:
        assert c.setInputBounds([-ub] * s["N"], [ub] * s["N"], (0, s["ch"]))
    return c


def _solve(c, X0, **kw):
    import torch
    r = c.optimizeBatch(torch.from_numpy(X0), torch.zeros(X0.shape[0], c.nu, dtype=torch.float64), **kw)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in r.items() if k != "_keep"}


def _attempts(c, B):
    """per instance (the statistics block's slot 14): attempts -- 1, or 2 where an instance that began with the inverse form failed and was solved again
    from the start with the factor (bit 1) -- and whether it left the inverse form on the way (bit 2: a sub-problem failed its check)"""
    w = np.array([int(c.debug_workspace(i)["scal"][14]) for i in range(B)])
    return 1 + ((w >> 1) & 1), (w >> 2) & 1


def _bound_rows(s, ub):
    """the kernel's bound table (nlmpc_capi.cpp sync_bounds): per decision variable the upper row, then the lower"""
    nxs = s["ph"] * 2 * s["N"]
    return [(nxs + k, sg) for k in range(s["ch"] * s["N"]) for sg in (1.0, -1.0)]


# (eight oscillators WITH their 240 bound rows fit the workgroup form's LDS block since round 6 -- 157.8 of 160 KB: plans in which no row reads a
# state keep neither the tables that say which nor an LDS copy of the multipliers -- so every variant runs for both shapes)
@pytest.mark.parametrize("shape,variant", [(sh, v) for sh in ("osc6", "osc8") for v in VARIANTS])
def test_bound_active_oscillators_match_the_golden_oracle_solutions(shape, variant, monkeypatch):
    s = SHAPES[shape]
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "nlmpc_oracle_solutions.json")))[s["key"]]
    assert len(gold["cases"]) >= 24 and sum(k["inputs_on_a_bound"] >= 0.75 * s["N"] * s["ch"] for k in gold["cases"]) >= 16
    worst = worst_cost = 0.0
    for ub in sorted({k["ub"] for k in gold["cases"]}):
        cases = [k for k in gold["cases"] if k["ub"] == ub]
        assert all(k["success"] for k in cases)
        c = _controller(monkeypatch, shape, ub, variant)
        X0 = np.array([k["x0"] for k in cases])
        r = _solve(c, X0, sequences=True)
        assert (r["status"] == 0).all(), (ub, r["solver_status"])
        assert (np.abs(r["seq_input"]) <= ub + 1e-9).all()
        for b, k in enumerate(cases):
            np.testing.assert_allclose(r["cmd"][b], k["cmd"], rtol=1e-5, atol=1e-5)
            worst = max(worst, np.abs(r["cmd"][b] - k["cmd"]).max() / max(1.0, np.abs(k["cmd"]).max()))
            worst_cost = max(worst_cost, abs(r["cost"][b] - k["cost"]) / abs(k["cost"]))
            assert abs(r["cost"][b] - k["cost"]) <= COST_RTOL * abs(k["cost"]), (b, r["cost"][b], k["cost"])
    print("parity %s with input bounds, %s: max |cmd - oracle| / max(1, |cmd|) = %.2e, max |cost - oracle| / cost = %.1e over %d golden instances"
          % (shape, variant, worst, worst_cost, len(gold["cases"])))


def _oracle_job(job):
    shape, x0, ub = job
    m = nlmpc_c.make(shape)
    N = m.nu
    o = m.solve(np.asarray(x0), np.zeros(N), max_iter=400, hard=True, lb_u=[-ub] * N, ub_u=[ub] * N)
    return dict(cmd=o["cmd"], cost=o["cost"], success=o["success"])


def _oracle_batch(shape, X0, ub, idx):
    """the oracle's SLSQP (compiled callbacks) on the instances idx, over the host's cores"""
    import multiprocessing as mp
    with mp.get_context("fork").Pool(max(1, min(16, (os.cpu_count() or 2) - 1))) as pool:
        return pool.map(_oracle_job, [(shape, X0[b], ub) for b in idx], chunksize=2)


def _kkt_at_point(m, z, x0, s, ub):
    """KKT conditions of the RESTATED problem (oracle callbacks) at z, independent of the kernel's multipliers: the multipliers of the dynamics
    (free) and of the active rows (>= 0: bounds within 1e-9, user rows within 1e-9 of zero) are the least-squares ones.  With every input on a
    bound the kernel's own multipliers are not usable for this: they satisfy B p + g_r + N' u = 0 with a curvature estimate B whose largest
    eigenvalue is 1e9 and more there (steps of 1e-7 along the pinned directions), so a converged step of 1e-7 leaves |B p| = O(|g|) in them.
    Returns (stationarity relative to |grad|_inf, primal violation)."""
    from scipy.optimize import lsq_linear
    m.x0 = np.asarray(x0, float)
    f, g = m.objective(z)
    c, Jc = m.state_eq(z)
    gi, Ji = m.user_ineq(z)
    nfree = m.nz - 1
    nxs = s["ph"] * 2 * s["N"]
    cols = [Jc[:, :nfree].T]
    nlam = Jc.shape[0]
    act = [Ji[k, :nfree] for k in range(gi.size) if gi[k] >= -1e-9]
    for k in range(s["ch"] * s["N"]):
        for sg in (1.0, -1.0):
            if abs(sg * z[nxs + k] - ub) <= 1e-9:
                e = np.zeros(nfree); e[nxs + k] = sg
                act.append(e)
    A = np.hstack(cols + ([np.array(act).T] if act else []))
    lo = np.concatenate([np.full(nlam, -np.inf), np.zeros(len(act))])
    sol = lsq_linear(A, -g[:nfree], bounds=(lo, np.full(A.shape[1], np.inf)), tol=1e-14)
    stat = np.abs(A @ sol.x + g[:nfree]).max() / max(1.0, np.abs(g).max())
    viol = max(np.abs(c).max(), gi.max(), (np.abs(z[nxs:-1]) - ub).max())
    return stat, viol


@pytest.mark.parametrize("shape,ub,B,variant", [("osc6", 0.05, 1024, "default"), ("osc6", 0.15, 1024, "default"), ("osc6", 0.05, 256, "wg-inverse-not-carried"),
                                                 ("osc6", 0.05, 256, "wg-factor"), ("osc6", 0.05, 256, "wave"), ("osc8", 0.05, 256, "default"), ("osc8", 0.15, 256, "default"), ("osc8", 0.05, 128, "wg-factor"), ("osc8", 0.05, 128, "wave")])
def test_every_bound_active_instance_solves_to_the_oracle_optimum(shape, ub, B, variant, monkeypatch):
    import torch
    s = SHAPES[shape]
    N = s["N"]
    rng = np.random.default_rng(2025 + N)
    X0 = rng.uniform(-0.5, 0.5, size=(B, 2 * N)); X0[:, 0] += 1.0                  # the distribution of tests/hunt_inconsistent_gpu.py
    c = _controller(monkeypatch, shape, ub, variant)
    r = _solve(c, X0, sequences=True)
    codes = {int(k): int((r["solver_status"] == k).sum()) for k in np.unique(r["solver_status"])}
    assert (r["status"] == 0).all(), codes                                         # round 5: {-4: 22, -3: 3, -1: 18} of 1024 here
    assert (np.abs(r["seq_input"]) <= ub).all()                                    # bounds hold exactly (the iterates are clamped, as nlopt's are)
    ev = c.evaluate(torch.from_numpy(r["z"]), torch.from_numpy(X0), grad=False, eq_jac=False, ineq_jac=False)
    assert ev["ceq"].abs().max().item() < 1e-7 and ev["cineq"].max().item() < 1e-9
    np.testing.assert_allclose(ev["cost"].cpu().numpy(), r["cost"], rtol=1e-12)
    on_bound = (np.abs(np.abs(r["z"][:, s["ph"] * 2 * N:-1]) - ub) <= 1e-9).sum(axis=1)
    left = _attempts(c, B) if c._lib.mpcx_nlmpc_last_form(c._h) > 0 else (np.ones(B, int), np.zeros(B, int))
    # against the oracle's SLSQP run here: six oscillators ~1 s per instance (the first 256 of the batch), eight ~10 s (a sample of 24)
    idx = list(range(min(B, 256))) if N == 6 else list(range(0, B, max(1, B // 24)))
    orc = _oracle_batch(shape, X0, ub, idx)
    worst = worst_cost = 0.0
    compared = 0
    for b, o in zip(idx, orc):
        if not o["success"]:
            continue                                     # (scipy's SLSQP gives up on a few starts: nothing to compare with)
        compared += 1
        np.testing.assert_allclose(r["cmd"][b], o["cmd"], rtol=1e-5, atol=1e-5)
        assert abs(r["cost"][b] - o["cost"]) <= COST_RTOL * abs(o["cost"]), (b, r["cost"][b], o["cost"])
        worst = max(worst, np.abs(r["cmd"][b] - o["cmd"]).max() / max(1.0, np.abs(o["cmd"]).max()))
        worst_cost = max(worst_cost, abs(r["cost"][b] - o["cost"]) / abs(o["cost"]))
    assert compared >= 0.95 * len(idx)
    # ... and the restated problem's KKT conditions at the returned point on another sample (the part of the batch the oracle did not see included)
    mk = ref.oscillators(N=N, ph=s["ph"], ch=s["ch"], Ts=0.1)
    kk = np.zeros(2)
    for b in range(B - 1, 0, -(B // (12 if N == 6 else 4))):
        kk = np.maximum(kk, _kkt_at_point(mk, r["z"][b], X0[b], s, ub))
    assert kk[0] <= 2e-5 and kk[1] <= 1e-7, kk
    print("%s |u| <= %.2f, %s, %d instances: all solved (codes %s), inputs on a bound %d .. %d of %d; %d left the inverse form on the way, %d were solved again from the start; "
          "%d against the oracle: max |cmd - oracle| / max(1, |cmd|) = %.2e, max |cost - oracle| / cost = %.1e; KKT at the point: stationarity %.1e violation %.1e"
          % (shape, ub, variant, B, codes, on_bound.min(), on_bound.max(), s["ch"] * N, left[1].sum(), (left[0] == 2).sum(), compared, worst, worst_cost, kk[0], kk[1]))


def test_instances_that_leave_the_inverse_form_end_where_the_factor_form_ends(monkeypatch):
    """Six oscillators, |u| <= 0.05, 1024 instances: the default plan keeps the Schur complement's inverse, checks every sub-problem's working
    rows, and solves the first sub-problem that fails the check -- and all that follow -- with the factor.  A fifth of these instances do
    (counted: the statistics block); every one of the 1024 ends at the optimum a handle planned without the inverse form ends at.
    (Round 5 took the inverse form's results unchecked: 43 .. 51 of these 1024 ended with status ERROR, and of 256 two more with status
    SUCCESS at a point whose cost was 2e-4 and 5e-4 above the optimum.)"""
    N, B, ub = 6, 1024, 0.05
    rng = np.random.default_rng(2025 + N)
    X0 = rng.uniform(-0.5, 0.5, size=(B, 2 * N)); X0[:, 0] += 1.0
    c = _controller(monkeypatch, "osc6", ub, "default")
    a = _solve(c, X0)
    att, left = _attempts(c, B)
    f = _controller(monkeypatch, "osc6", ub, "wg-factor")
    b = _solve(f, X0)
    fa, fl = _attempts(f, B)
    assert (fa == 1).all() and not fl.any()
    assert (a["status"] == 0).all() and (b["status"] == 0).all()
    print("six oscillators, |u| <= 0.05: %d of %d instances left the inverse form on the way, %d were solved again from the start" % (left.sum(), B, (att == 2).sum()))
    assert left.sum() >= 1                                   # the path this test is about is exercised
    np.testing.assert_allclose(a["cmd"], b["cmd"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(a["cost"], b["cost"], rtol=COST_RTOL)
    # an instance that was solved again from the start (none here today) is the factor form's result bit for bit
    again = att == 2
    for key in ("cmd", "cost", "z", "iterations", "solver_status"):
        assert np.array_equal(a[key][again], b[key][again]), key
