"""-m gpu: the NLMPC transcription kernels (mpcx_nlmpc_evaluate_batch) against the oracle's restatement of
Mapping / Objective / Constraints (oracle/nlmpc_numpy.py), on the reference's example systems."""
import numpy as np
import pytest

from oracle import nlmpc_numpy as ref

pytestmark = pytest.mark.gpu


def _cases():
    return [("vanderpol", dict(ph=10, ch=5, Ts=0.1)), ("vanderpol", dict(ph=10, ch=10, Ts=0.1)),
            ("vanderpol", dict(ph=7, ch=3, Ts=0.05)), ("ugv", dict(ph=30, ch=30)), ("ugv", dict(ph=12, ch=4)),
            ("osc6", dict(ph=20, ch=10, Ts=0.1)), ("osc8", dict(ph=6, ch=3, Ts=0.1))]


def _make(name, kw):
    from libmpc_amd.nlmpc import NLMPCEvaluator, VANDERPOL, UGV, OSCILLATORS6, OSCILLATORS8
    if name == "vanderpol":
        return ref.vanderpol(**kw), NLMPCEvaluator(VANDERPOL, kw["ph"], kw["ch"], kw["Ts"])
    if name.startswith("osc"):
        n = int(name[3:])
        return ref.oscillators(N=n, **kw), NLMPCEvaluator(OSCILLATORS6 if n == 6 else OSCILLATORS8, kw["ph"], kw["ch"], kw["Ts"])
    return ref.ugv(**kw), NLMPCEvaluator(UGV, kw["ph"], kw["ch"], 0.1)


@pytest.mark.parametrize("name,kw", _cases())
def test_transcription_matches_oracle(name, kw):
    import torch
    m, ev = _make(name, kw)
    assert (ev.nz, ev.neq, ev.nineq) == (m.nz, m.ph * m.nx, m.ineq)
    rng = np.random.default_rng(7)
    B = 9 if m.nz < 100 else 3
    Z = rng.normal(scale=1.5, size=(B, m.nz)); Z[:, -1] = rng.normal(scale=0.1, size=B)
    Z[0] = 0.0                                    # cold-start point: all finite-difference steps at their floor
    X0 = rng.normal(size=(B, m.nx))
    out = ev.evaluate(torch.from_numpy(Z), torch.from_numpy(X0))
    torch.cuda.synchronize()
    Jd = ev.dense_eq_jacobian(out["jeq"])
    o = {k: v.cpu().numpy() for k, v in out.items()}
    for b in range(B):
        m.x0 = X0[b]
        f0, g = m.objective(Z[b])
        c, J = m.state_eq(Z[b])
        gi, Ji = m.user_ineq(Z[b])
        # values: same arithmetic, fp64 round-off only
        assert abs(o["cost"][b] - f0) <= 1e-12 * max(1.0, abs(f0))
        np.testing.assert_allclose(o["ceq"][b], c, rtol=0, atol=1e-13 * max(1.0, np.abs(c).max()))
        np.testing.assert_allclose(o["cineq"][b], gi, rtol=0, atol=1e-13)
        # finite differences amplify round-off by 1/step (1.5e-8): eps*|f|/step
        tol_g = 64 * np.finfo(float).eps * max(1.0, abs(f0)) / ref.DV
        np.testing.assert_allclose(o["grad"][b], g, rtol=1e-9, atol=tol_g)
        np.testing.assert_allclose(Jd[b], J, rtol=1e-9, atol=1e-6)
        np.testing.assert_allclose(o["jineq"][b], Ji, rtol=1e-9, atol=1e-6)


def test_equality_jacobian_is_the_derivative():
    """size-independent property at the full horizon: J_eq (z2 - z1) predicts ceq(z2) - ceq(z1) to second order"""
    import torch
    m, ev = _make("ugv", dict(ph=30, ch=30))
    rng = np.random.default_rng(3)
    B = 256
    Z = rng.normal(size=(B, m.nz)); D = 1e-4 * rng.normal(size=(B, m.nz)); X0 = rng.normal(size=(B, m.nx))
    a = ev.evaluate(torch.from_numpy(Z), torch.from_numpy(X0), cost=False, grad=False, ineq_jac=False)
    b = ev.evaluate(torch.from_numpy(Z + D), torch.from_numpy(X0), cost=False, grad=False, eq_jac=False, ineq_jac=False)
    J = ev.dense_eq_jacobian(a["jeq"])
    pred = np.einsum("brc,bc->br", J, D)
    np.testing.assert_allclose((b["ceq"] - a["ceq"]).cpu().numpy(), pred, atol=1e-9)   # the UGV dynamics are linear


def test_partial_outputs_and_empty_batch():
    import torch
    m, ev = _make("vanderpol", dict(ph=10, ch=5, Ts=0.1))
    z = torch.zeros(3, m.nz, dtype=torch.float64); x0 = torch.ones(3, m.nx, dtype=torch.float64)
    o = ev.evaluate(z, x0, grad=False, eq_jac=False, ineq_jac=False)
    assert o["grad"] is None and o["jeq"] is None and o["cost"].shape == (3,)
    m.x0 = np.ones(2)
    assert abs(o["cost"][0].item() - m.objective(np.zeros(m.nz), False)[0]) < 1e-12
    o = ev.evaluate(z[:0], x0[:0])
    assert o["cost"].shape == (0,)


# ---- the SQP solve (rows a21-a23) --------------------------------------------------------------------------------
# Tolerance: both sides differentiate by finite differences (forward, step 1.5e-8) and stall at that noise floor; the
# oracle (scipy SLSQP on the restated callbacks) itself stops with "positive directional derivative" there.  u* is
# compared at north_star's 1e-5 relative (of max(1, |u*|); measured 3e-7 ... 1e-6, printed by the tests), the optimal cost at
# 1e-8 relative.
def _solve_case(name, kw, X0, U0, hard, max_iter):
    import torch
    from libmpc_amd.nlmpc import NLMPC, NLParameters, VANDERPOL, UGV, OSCILLATORS6
    c = NLMPC(dict(vanderpol=VANDERPOL, ugv=UGV, osc6=OSCILLATORS6)[name], kw["ph"], kw["ch"], kw.get("Ts", 0.1))
    c.setOptimizerParameters(NLParameters(maximum_iteration=max_iter, hard_constraints=int(hard)))
    r = c.optimizeBatch(torch.from_numpy(X0), torch.from_numpy(U0), sequences=True)
    torch.cuda.synchronize()
    return c, {k: v.cpu().numpy() for k, v in r.items() if k != "_keep"}


def test_vanderpol_solve_matches_oracle():
    kw = dict(ph=10, ch=5, Ts=0.1)
    rng = np.random.default_rng(11)
    B = 24
    X0 = rng.uniform(-1.0, 1.0, size=(B, 2)); X0[0] = [0.0, 1.0]          # examples/vanderpol_ex.cpp:67
    U0 = np.zeros((B, 1))
    c, r = _solve_case("vanderpol", kw, X0, U0, True, 200)
    assert (r["status"] == 0).all(), (r["status"], r["solver_status"], r["iterations"])
    m = ref.vanderpol(**kw)
    compared = 0
    worst = 0.0
    for b in range(B):
        o = m.solve(X0[b], U0[b], max_iter=1000)
        if not o["success"]:            # scipy's SLSQP gives up on some starts (infeasible end point): nothing to compare with
            assert np.abs(m.state_eq(o["z"], False)[0]).max() > 1e-6
            continue
        compared += 1
        assert abs(r["cost"][b] - o["cost"]) <= 1e-8 * max(1.0, abs(o["cost"]))
        worst = max(worst, np.abs(r["cmd"][b] - o["cmd"]).max() / max(1.0, np.abs(o["cmd"]).max()))
        np.testing.assert_allclose(r["cmd"][b], o["cmd"], rtol=1e-5, atol=1e-5)       # north_star: 1e-5 relative (of max(1, |u*|))
        np.testing.assert_allclose(r["seq_state"][b], o["X"], atol=2e-5)
        assert r["is_feasible"][b] == 1 and (r["seq_input"][b] <= 0.5 + 1e-9).all()
    assert compared >= B - 2
    print("parity vanderpol (config 1): max |cmd - oracle| / max(1, |cmd|) = %.2e over %d instances" % (worst, compared))
    assert not r["seq_output"].any()                  # no output function in the example: zeros (Model.hpp:72-96)


def test_ugv_solve_matches_oracle():
    kw = dict(ph=30, ch=30)
    rng = np.random.default_rng(1)
    B = 6
    X0 = np.zeros((B, 4)); X0[1:, :2] = rng.uniform(-0.5, 0.5, size=(B - 1, 2))
    U0 = np.zeros((B, 2))
    c, r = _solve_case("ugv", kw, X0, U0, False, 150)
    m = ref.ugv(**kw)
    assert (r["status"] != 3).all(), (r["status"], r["solver_status"], r["iterations"])
    assert c.ny == 4 and np.array_equal(r["seq_output"], r["seq_state"])       # y = C x with C = I (ugv_ex.cpp:34-77)
    worst = 0.0
    for b in range(B):
        o = m.solve(X0[b], U0[b], max_iter=100, hard=False)
        assert abs(r["cost"][b] - o["cost"]) <= 1e-7 * abs(o["cost"]), (b, r["cost"][b], o["cost"])
        ocmd = o["cmd"]
        if b == 0 and not np.allclose(r["cmd"][b], ocmd, rtol=1e-5, atol=1e-5):
            # the example's own start x0 = 0: two paths round the obstacles, mirror images of each other in the command, whose costs agree
            # to 2e-8 (asserted above) -- round-off decides which one a solver takes
            ocmd = ocmd[::-1]
        worst = max(worst, np.abs(r["cmd"][b] - ocmd).max() / max(1.0, np.abs(ocmd).max()))
        np.testing.assert_allclose(r["cmd"][b], ocmd, rtol=1e-5, atol=1e-5)
    print("parity ugv (config 3): max |cmd - oracle| / max(1, |cmd|) = %.2e over %d instances" % (worst, B))


def test_solution_satisfies_kkt_at_scale():
    """size-independent properties on a full batch: dynamics defects ~ 0, inequalities respected, cost not above cold start"""
    import torch
    kw = dict(ph=30, ch=30)
    rng = np.random.default_rng(5)
    B = 4096                                                     # BASELINE config 3's batch
    X0 = np.zeros((B, 4)); X0[:, :2] = rng.uniform(-0.5, 0.5, size=(B, 2))
    U0 = np.zeros((B, 2))
    c, r = _solve_case("ugv", kw, X0, U0, False, 150)
    ok = r["status"] != 3
    assert ok.mean() > 0.98, ok.mean()
    ev = c.evaluate(torch.from_numpy(r["z"]), torch.from_numpy(X0), grad=False, eq_jac=False, ineq_jac=False)
    assert ev["ceq"].abs().max().item() < 1e-7
    assert ev["cineq"][torch.from_numpy(ok).cuda()].max().item() < 1e-7
    z0 = np.concatenate([np.tile(X0, (1, 30)), np.zeros((B, 61))], axis=1)
    f0 = c.evaluate(torch.from_numpy(z0), torch.from_numpy(X0), grad=False, eq=False, eq_jac=False, ineq=False, ineq_jac=False)["cost"]
    assert (ev["cost"] <= f0)[torch.from_numpy(ok).cuda()].all()


def test_warm_start_reproduces_and_is_cheaper():
    import torch
    from libmpc_amd.nlmpc import NLMPC, NLParameters, VANDERPOL
    c = NLMPC(VANDERPOL, 10, 5, 0.1)
    c.setOptimizerParameters(NLParameters(maximum_iteration=200))
    x0 = torch.tensor([[0.0, 1.0]] * 4, dtype=torch.float64); u0 = torch.zeros(4, 1, dtype=torch.float64)
    a = c.optimizeBatch(x0, u0)
    # plant step as in the example's closed loop (vanderpol_ex.cpp:76-85), then re-solve from the shifted solution
    x = x0.cuda(); u = a["cmd"]
    dx = torch.stack([(1 - x[:, 1] ** 2) * x[:, 0] - x[:, 1] + u[:, 0], x[:, 0]], dim=1)
    x1 = x + 0.1 * dx
    cold = c.optimizeBatch(x1, u)
    warm = c.optimizeBatch(x1, u, z_warm=a["z"])
    torch.cuda.synchronize()
    assert (warm["status"] == 0).all() and (cold["status"] == 0).all()
    np.testing.assert_allclose(warm["cmd"].cpu().numpy(), cold["cmd"].cpu().numpy(), rtol=2e-5, atol=2e-6)
    assert warm["iterations"].float().mean().item() <= cold["iterations"].float().mean().item()


def test_oscillator_network_solve_matches_oracle():
    """examples/networked_oscillators_ex.cpp: x0 = e_0 (one oscillator displaced), and two perturbed starts (config 5)"""
    kw = dict(ph=20, ch=10, Ts=0.1)
    rng = np.random.default_rng(2)
    B = 3
    X0 = rng.uniform(-0.1, 0.1, size=(B, 12)); X0[:, 0] += 1.0; X0[0] = 0.0; X0[0, 0] = 1.0
    U0 = np.zeros((B, 6))
    c, r = _solve_case("osc6", kw, X0, U0, True, 200)
    assert (r["status"] == 0).all(), (r["status"], r["solver_status"], r["iterations"])
    m = ref.oscillators(N=6, **kw)
    worst = 0.0
    for b in range(B):
        o = m.solve(X0[b], U0[b], max_iter=200)
        assert o["success"], o["message"]
        assert abs(r["cost"][b] - o["cost"]) <= 1e-8 * max(1.0, abs(o["cost"]))
        worst = max(worst, np.abs(r["cmd"][b] - o["cmd"]).max() / max(1.0, np.abs(o["cmd"]).max()))
        np.testing.assert_allclose(r["cmd"][b], o["cmd"], rtol=1e-5, atol=1e-5)
    print("parity 6 oscillators: max |cmd - oracle| / max(1, |cmd|) = %.2e over %d instances" % (worst, B))


def test_state_and_input_bounds_match_oracle():
    """NLMPC::setInputBounds / setStateBounds (NLMPC.hpp:285-398 -> nlopt bounds): as rows of the sub-problem"""
    import torch
    from libmpc_amd.nlmpc import NLMPC, NLParameters, VANDERPOL
    kw = dict(ph=10, ch=5, Ts=0.1)
    c = NLMPC(VANDERPOL, 10, 5, 0.1)
    c.setOptimizerParameters(NLParameters(maximum_iteration=200))
    assert c.setInputBounds([-0.3], [0.3], (0, 5))
    assert c.setStateBounds([-0.8, -2.0], [0.8, 2.0], (-1, -1))
    assert not c.setStateBounds([-1, -1], [1, 1], (3, 2)) and not c.setInputBounds([-1], [1], (0, 6))
    with pytest.raises(RuntimeError):
        c.setOutputBounds([0, 0], [1, 1])
    rng = np.random.default_rng(21)
    B = 12
    X0 = rng.uniform(-0.7, 0.7, size=(B, 2)); X0[0] = [0.0, 1.0]
    U0 = np.zeros((B, 1))
    r = c.optimizeBatch(torch.from_numpy(X0), torch.from_numpy(U0), sequences=True)
    torch.cuda.synchronize()
    r = {k: v.cpu().numpy() for k, v in r.items() if k != "_keep"}
    m = ref.vanderpol(**kw)
    compared = failed_both = 0
    for b in range(B):
        o = m.solve(X0[b], U0[b], max_iter=1000, lb_u=[-0.3], ub_u=[0.3], lb_x=[-0.8, -2.0], ub_x=[0.8, 2.0])
        if not o["success"]:
            # some starts cannot stay inside |x_0| <= 0.8 with |u| <= 0.3: the problem has no feasible point.  Kraft's SLSQP (with its
            # relaxed LSEI sub-problem) ends these in mode 8, "positive directional derivative for linesearch", at an infeasible point;
            # NLopt's translation turns that into NLOPT_ROUNDOFF_LIMITED, its C++ wrapper throws, and NLOptimizer::run
            # (NLOptimizer.hpp:561-570, 611-617) reports ERROR with the previous command and an infinite cost.  The kernel finds the
            # linearised rows inconsistent and reports exactly that: ERROR, solver status -1, cmd = lastU, cost = inf.
            assert "Positive directional derivative" in o["message"] and np.abs(m.state_eq(o["z"], False)[0]).max() > 1e-4
            assert r["status"][b] == 3 and r["solver_status"][b] == -1 and r["cmd"][b, 0] == U0[b, 0] and np.isinf(r["cost"][b])
            failed_both += 1
            continue
        assert r["status"][b] == 0, (b, r["solver_status"][b])
        assert (r["seq_input"][b] <= 0.3 + 1e-9).all() and (r["seq_input"][b] >= -0.3 - 1e-9).all()
        assert (np.abs(r["seq_state"][b][1:, 0]) <= 0.8 + 1e-9).all()
        compared += 1
        assert abs(r["cost"][b] - o["cost"]) <= 1e-8 * max(1.0, abs(o["cost"])), (b, r["cost"][b], o["cost"])
        np.testing.assert_allclose(r["cmd"][b], o["cmd"], rtol=1e-5, atol=1e-5)
    assert compared >= B - 3 and compared + failed_both == B
    print("bounds: %d starts compared with the oracle, %d infeasible problems reported as ERROR by both" % (compared, failed_both))
    # matrix form, one column per step; without the state bounds every start is feasible
    assert c.setInputBounds(np.full((1, 5), -0.2), np.full((1, 5), 0.2))
    assert c.setStateBounds([-np.inf, -np.inf], [np.inf, np.inf], (-1, -1))
    r2 = c.optimizeBatch(torch.from_numpy(X0), torch.from_numpy(U0), sequences=True)
    torch.cuda.synchronize()
    assert (r2["status"] == 0).all() and (r2["seq_input"].abs() <= 0.2 + 1e-9).all()


def _golden(key):
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "nlmpc_oracle_solutions.json")))[key]


def _compare_with_golden(c, gold, label, cost_rtol, oracle_model=None, hard=True):
    """every golden instance in one batch; compared wherever the oracle's SLSQP ended at a usable point.  Returns (compared, other):
    an instance where the two solvers stopped at different local optima is counted (with how its cost compares), not compared -- and one
    that ended WORSE than the oracle's has to be a KKT point of the restated problem all the same (oracle callbacks, the kernel's multipliers)"""
    import torch
    X0 = np.array([k["x0"] for k in gold["cases"]]); U0 = np.array([k["u0"] for k in gold["cases"]])
    r = c.optimizeBatch(torch.from_numpy(X0), torch.from_numpy(U0), multipliers=True)
    torch.cuda.synchronize()
    status, cost, cmd = r["status"].cpu().numpy(), r["cost"].cpu().numpy(), r["cmd"].cpu().numpy()
    zs, mus = r["z"].cpu().numpy(), r["multipliers"].cpu().numpy()
    worst, worst_cost, compared, other, bad_cost = 0.0, 0.0, 0, [], []
    # usable: scipy's SLSQP converged, or ended in its mode 8 (line search cannot improve: the step is below what its ftol of 1e-12
    # resolves) at a feasible point -- how it ends most UGV solves.  Its diverged runs (cost 1e9 and more on config 5) are not.
    usable = [k["success"] or (k["slsqp_mode"] == 8 and k["eq_violation"] < 1e-8 and k["ineq_violation"] < 1e-6) for k in gold["cases"]]
    for b, k in enumerate(gold["cases"]):
        if not usable[b]:
            continue
        assert status[b] != 3, (b, status[b])
        if not np.allclose(cmd[b], k["cmd"], rtol=1e-5, atol=1e-5):
            # another local optimum: the UGV's obstacle rows make the problem non-convex (from the example's own start x0 = 0 the two
            # mirror-image paths cost the same to 1e-9 and round-off decides which one a solver takes; from other starts the two SLSQP
            # implementations pass the obstacles on different sides).  Counted, not compared: the caller bounds how many there may be.
            assert status[b] == 0, (b, status[b])
            how = "better" if cost[b] < k["cost"] * (1.0 - cost_rtol) else ("equal" if cost[b] <= k["cost"] * (1.0 + cost_rtol) else "worse")
            if how == "worse" and oracle_model is not None:
                stat, viol, comp, neg = _kkt_report(oracle_model, zs[b], X0[b], mus[b], hard)
                assert stat <= 2e-5 and viol <= 1e-7 and comp <= 1e-5 and neg >= -1e-7, (b, stat, viol, comp, neg)
            other.append((b, how))
            continue
        if abs(cost[b] - k["cost"]) > cost_rtol * abs(k["cost"]):
            bad_cost.append((b, cost[b], k["cost"], k["slsqp_mode"], k["ineq_violation"]))
        worst_cost = max(worst_cost, abs(cost[b] - k["cost"]) / abs(k["cost"]))
        compared += 1
        worst = max(worst, np.abs(cmd[b] - k["cmd"]).max() / max(1.0, np.abs(k["cmd"]).max()))
    print("parity %s: max |cmd - oracle| / max(1, |cmd|) = %.2e, max |cost - oracle| / cost = %.1e over %d golden instances (%d usable end points "
          "of the oracle; %d at another local optimum: %s)" % (label, worst, worst_cost, compared, sum(usable), len(other),
             {w: sum(1 for _, x in other if x == w) for w in ("better", "equal", "worse")}))
    assert not bad_cost, bad_cost
    return compared, other


def _force_form(monkeypatch, form):
    """the kernel form of the handles created next: "default" = the library's choice, "wg" / "wave" forced (read at create)"""
    for k in ("MPCX_NLMPC_FORM", "MPCX_NLMPC_WAVES", "MPCX_NLMPC_BLOCKS"):
        monkeypatch.delenv(k, raising=False)
    if form != "default":
        monkeypatch.setenv("MPCX_NLMPC_FORM", form)


@pytest.mark.parametrize("form", ["default", "wg", "wave"])
def test_eight_oscillator_config_matches_golden_oracle_solutions(form, monkeypatch):
    """BASELINE config 5 (nz = 601, 480 equalities, 248 inequalities): the oracle needs 60 ... 90 s per instance, its solutions on
    the example's start and the first instances of bench.py's batch are kept in tests/golden/nlmpc_oracle_solutions.json
    (generated by tests/golden/make_nlmpc_golden.py).  Every form of the kernel against all 64 of them -- whichever one the
    library picks at the quoted batch is among them."""
    from libmpc_amd.nlmpc import NLMPC, NLParameters, OSCILLATORS8
    _force_form(monkeypatch, form)
    gold = _golden("oscillators8_ph30_ch15")
    assert len(gold["cases"]) >= 64
    c = NLMPC(OSCILLATORS8, gold["ph"], gold["ch"], gold["Ts"])
    c.setOptimizerParameters(NLParameters(maximum_iteration=200))
    compared, other = _compare_with_golden(c, gold, "8 oscillators (config 5), form %s" % form, 1e-8)
    assert compared >= 56 and not other                # one optimum here; scipy's SLSQP diverges on a few of the 64 starts


@pytest.mark.parametrize("form", ["default", "wg", "wave"])
def test_ugv_config_matches_golden_oracle_solutions(form, monkeypatch):
    """BASELINE config 3 (ugv_ex.cpp, soft constraints): the example's start and the first 255 instances of bench.py's batch, every form of
    the kernel against all 256"""
    from libmpc_amd.nlmpc import NLMPC, NLParameters, UGV
    _force_form(monkeypatch, form)
    gold = _golden("ugv_ph30_ch30")
    assert len(gold["cases"]) >= 256
    c = NLMPC(UGV, gold["ph"], gold["ch"], gold["Ts"])
    c.setOptimizerParameters(NLParameters(maximum_iteration=150, hard_constraints=0))
    # cost: scipy's mode-8 end points sit up to 1e-7 outside the obstacle rows (stored: ineq_violation), which buys them up to 5e-7 of cost
    # measured: 224 of the 231 usable end points agree, 7 sit at another local optimum (1 of equal cost, 6 worse -- each of them a KKT point
    # of the restated problem, checked inside); the allowance is that count + 2
    compared, other = _compare_with_golden(c, gold, "ugv (config 3), form %s" % form, 2e-6, oracle_model=ref.ugv(ph=gold["ph"], ch=gold["ch"]), hard=False)
    assert compared >= 218 and len(other) <= 9, (compared, other)


@pytest.mark.parametrize("B", [256, 1024])
def test_config5_properties_and_kkt_at_batch_256(B):
    """BASELINE config 5 (8 oscillators, ph 30, ch 15) on a batch of 256 and on the per-GPU batch the config quotes (1024): every instance converges, the returned points are
    feasible to round-off, and on a sample the restated problem's KKT conditions hold with
    the kernel's multipliers (oracle callbacks only -- no scipy solve, which takes ~50 s per instance here)"""
    import torch
    from libmpc_amd.nlmpc import NLMPC, NLParameters, OSCILLATORS8
    kw = dict(ph=30, ch=15, Ts=0.1)
    rng = np.random.default_rng(0)
    X0 = rng.uniform(-0.1, 0.1, size=(B, 16)); X0[:, 0] += 1.0
    U0 = np.zeros((B, 8))
    c = NLMPC(OSCILLATORS8, 30, 15, 0.1)
    c.setOptimizerParameters(NLParameters(maximum_iteration=200))
    r = c.optimizeBatch(torch.from_numpy(X0), torch.from_numpy(U0), multipliers=True)
    torch.cuda.synchronize()
    assert (r["status"] == 0).all(), (r["status"].cpu().numpy(), r["solver_status"].cpu().numpy())
    ev = c.evaluate(r["z"], torch.from_numpy(X0), grad=False, eq_jac=False, ineq_jac=False)
    assert ev["ceq"].abs().max().item() < 1e-7
    assert ev["cineq"].max().item() < 1e-9
    np.testing.assert_allclose(ev["cost"].cpu().numpy(), r["cost"].cpu().numpy(), rtol=1e-12)      # the reported cost is the cost there
    m = ref.oscillators(N=8, **kw)
    z = r["z"].cpu().numpy(); mu = r["multipliers"].cpu().numpy()
    worst = np.zeros(4)
    for b in range(0, B, 64):
        stat, viol, comp, neg = _kkt_report(m, z[b], X0[b], mu[b], True)
        worst = np.maximum(worst, [stat, viol, comp, -neg])
    print("KKT config 5 (B=%d, sampled every 64th): stationarity %.2e  violation %.2e  complementarity %.2e  most negative multiplier %.2e"
          % (B, worst[0], worst[1], worst[2], -worst[3]))
    assert worst[0] <= 1e-4 and worst[1] <= 1e-7 and worst[2] <= 1e-6 and worst[3] <= 1e-9, worst


def test_config3_shards_equal_rows_of_the_unsharded_solve_across_the_residency_threshold():
    """BASELINE config 3 at its quoted batch: 4096 instances as one batch (blocks and reduced rows in the workspace, four workgroups per CU)
    and as shards of 512 (what eight ranks would each be given: resident with the blocks in LDS) -- another place for the same
    arithmetic: the shards are bit for bit the rows of the whole"""
    import torch
    from libmpc_amd.nlmpc import NLMPC, NLParameters, UGV
    m = NLMPC(UGV, 30, 30, 0.1)
    m.setOptimizerParameters(NLParameters(maximum_iteration=150, hard_constraints=0))
    rng = np.random.default_rng(0)
    Bs, R = 512, 8
    X0 = np.zeros((Bs * R, 4)); X0[:, :2] = rng.uniform(-0.5, 0.5, size=(Bs * R, 2))
    U0 = np.zeros((Bs * R, 2))
    whole = m.optimizeBatch(torch.from_numpy(X0), torch.from_numpy(U0)); torch.cuda.synchronize()
    assert (whole["status"] == 0).float().mean().item() >= 0.999
    for rk in (0, 3, 7):
        sl = slice(rk * Bs, (rk + 1) * Bs)
        part = m.optimizeBatch(torch.from_numpy(X0[sl]), torch.from_numpy(U0[sl])); torch.cuda.synchronize()
        for key in ("cmd", "cost", "status", "z", "iterations"):
            assert torch.equal(part[key], whole[key][sl]), (rk, key)


def test_user_equality_constraints_match_oracle():
    """a19: NLMPC::setEqConFunction -- Constraints::evaluateEq / computeEqJacobian and the equalities in the solve, on the
    Van der Pol system with the terminal constraint x(ph) = 0"""
    import torch
    from libmpc_amd.nlmpc import NLMPC, NLParameters, VANDERPOL_TERMINAL
    for ch in (10, 5):
        m = ref.vanderpol_terminal(ph=10, ch=ch)
        c = NLMPC(VANDERPOL_TERMINAL, 10, ch, 0.1)
        assert c.neq_user == 2 and c.nineq == 11
        rng = np.random.default_rng(13)
        B = 6
        Z = rng.normal(size=(B, m.nz)); X0 = rng.normal(size=(B, 2))
        ev = c.evaluate(torch.from_numpy(Z), torch.from_numpy(X0))
        torch.cuda.synchronize()
        for b in range(B):
            m.x0 = X0[b]
            h, Jh = m.user_eq(Z[b])
            np.testing.assert_allclose(ev["cineq"][b, 11:].cpu().numpy(), h, rtol=0, atol=1e-14)
            np.testing.assert_allclose(ev["jineq"][b, 11:].cpu().numpy(), Jh, rtol=1e-9, atol=1e-6)
            gi, Ji = m.user_ineq(Z[b])
            np.testing.assert_allclose(ev["jineq"][b, :11].cpu().numpy(), Ji, rtol=1e-9, atol=1e-6)
        c.setOptimizerParameters(NLParameters(maximum_iteration=300))
        X0 = rng.uniform(-0.12, 0.12, size=(B, 2)); X0[0] = [0.1, 0.1]
        U0 = np.zeros((B, 1))
        r = c.optimizeBatch(torch.from_numpy(X0), torch.from_numpy(U0), sequences=True)
        torch.cuda.synchronize()
        r = {k: v.cpu().numpy() for k, v in r.items() if k != "_keep"}
        compared = 0
        worst = 0.0
        for b in range(B):
            o = m.solve(X0[b], U0[b], max_iter=500)
            if not o["success"]:
                continue
            compared += 1
            assert r["status"][b] == 0 and r["is_feasible"][b] == 1, (b, r["solver_status"][b])
            assert np.abs(r["seq_state"][b][10]).max() <= 1e-9
            assert abs(r["cost"][b] - o["cost"]) <= 1e-7 * max(1.0, abs(o["cost"])), (b, r["cost"][b], o["cost"])
            worst = max(worst, np.abs(r["cmd"][b] - o["cmd"]).max() / max(1.0, np.abs(o["cmd"]).max()))
            np.testing.assert_allclose(r["cmd"][b], o["cmd"], rtol=1e-5, atol=1e-5)
        print("parity user equalities: max |cmd - oracle| / max(1, |cmd|) = %.2e over %d instances" % (worst, compared))
        assert compared >= B - 1


@pytest.mark.parametrize("name,kw,hard", [("ugv", dict(ph=12, ch=4), False), ("vanderpol", dict(ph=10, ch=10, Ts=0.1), True),
                                           ("vanderpol", dict(ph=7, ch=3, Ts=0.05), True)])
def test_move_blocking_variants_solve_like_the_oracle(name, kw, hard):
    """Mapping::computeMapping (Mapping.hpp:221-257): ch < ph holds the last move, ch = ph does not block at all"""
    rng = np.random.default_rng(31)
    B = 5
    nx = 4 if name == "ugv" else 2
    X0 = np.zeros((B, nx)); X0[:, :2] = rng.uniform(-0.4, 0.4, size=(B, 2))
    U0 = np.zeros((B, 2 if name == "ugv" else 1))
    c, r = _solve_case(name, kw, X0, U0, hard, 200)
    m = ref.ugv(**kw) if name == "ugv" else ref.vanderpol(**kw)
    compared = 0
    worst = 0.0
    for b in range(B):
        o = m.solve(X0[b], U0[b], max_iter=300, hard=hard)
        if not o["success"] and "Positive directional" not in o["message"]:
            continue
        if np.abs(m.state_eq(o["z"], False)[0]).max() > 1e-8:
            continue
        compared += 1
        assert r["status"][b] != 3, (b, r["solver_status"][b])
        assert abs(r["cost"][b] - o["cost"]) <= 1e-7 * max(1.0, abs(o["cost"])), (b, r["cost"][b], o["cost"])
        worst = max(worst, np.abs(r["cmd"][b] - o["cmd"]).max() / max(1.0, np.abs(o["cmd"]).max()))
        np.testing.assert_allclose(r["cmd"][b], o["cmd"], rtol=1e-5, atol=1e-5)
        # the held last block: every step from ch-1 on applies the same input
        assert np.abs(r["seq_input"][b][kw["ch"] - 1:] - r["seq_input"][b][kw["ch"] - 1]).max() == 0.0
    print("parity move blocking %s %s: max |cmd - oracle| / max(1, |cmd|) = %.2e over %d instances" % (name, kw, worst, compared))
    assert compared >= B - 1


def test_receding_horizon_with_carried_curvature_reaches_the_same_optimum():
    """extension: the next tick starts from the shifted solution (reference behaviour) and, optionally, from the curvature
    estimate the previous solve left behind -- fewer iterations, same optimum as a cold solve.  On the Van der Pol loop of
    examples/vanderpol_ex.cpp:76-85 (one optimum; the UGV's obstacles make the answer depend on the starting point)."""
    import torch
    from libmpc_amd.nlmpc import NLMPC, NLParameters, VANDERPOL
    B = 64
    rng = np.random.default_rng(17)
    x = torch.from_numpy(rng.uniform(-0.8, 0.8, size=(B, 2))).cuda(); u = torch.zeros(B, 1, dtype=torch.float64).cuda()
    warm = NLMPC(VANDERPOL, 10, 5, 0.1); cold = NLMPC(VANDERPOL, 10, 5, 0.1)
    for c in (warm, cold):
        c.setOptimizerParameters(NLParameters(maximum_iteration=200))
    z = None
    it_w, it_c = [], []
    for tick in range(5):
        rw = warm.optimizeBatch(x, u, z_warm=z, warm_curvature=True)
        rc = cold.optimizeBatch(x, u)
        torch.cuda.synchronize()
        ok = (rc["status"] == 0) & (rw["status"] == 0)
        assert ok.float().mean().item() > 0.95
        dc = (rw["cost"] - rc["cost"]).abs() / rc["cost"].abs().clamp_min(1.0)
        assert dc[ok].max().item() <= 1e-8, (tick, dc[ok].max().item())
        du = (rw["cmd"] - rc["cmd"]).abs().max(dim=1).values
        assert du[ok].max().item() <= 2e-5, (tick, du[ok].max().item())
        it_w.append(rw["iterations"].float().mean().item()); it_c.append(rc["iterations"].float().mean().item())
        u = rc["cmd"]; z = rw["z"]
        dx = torch.stack([(1 - x[:, 1] ** 2) * x[:, 0] - x[:, 1] + u[:, 0], x[:, 0]], dim=1)
        x = x + 0.1 * dx
    assert it_w[0] == it_c[0] and sum(it_w[1:]) < 0.8 * sum(it_c[1:]), (it_w, it_c)


# ---- oracle-independent optimality check, mapping scalings, nlopt's tolerances ------------------------------------------
def _kkt_report(m, z, x0, mu, hard, nbnd_rows=None):
    """KKT residuals of the RESTATED problem (oracle callbacks) at z with the kernel's inequality / equality / bound
    multipliers mu; the dynamics multipliers are the least-squares ones.  Returns (stationarity relative to |grad|_inf,
    primal violation, worst complementarity product, most negative inequality multiplier)."""
    m.x0 = np.asarray(x0, float)
    f, g = m.objective(z)
    c, Jc = m.state_eq(z)
    nfree = m.nz - 1 if hard else m.nz               # slack pinned at zero under hard constraints (NLOptimizer.hpp:182-186)
    rhs = g.copy()
    viol = np.abs(c).max()
    comp, neg, k = 0.0, 0.0, 0
    if m.ineq_fun is not None:
        gi, Ji = m.user_ineq(z)
        mi = mu[k:k + gi.size]; k += gi.size
        rhs += Ji.T @ mi
        viol = max(viol, gi.max())
        comp = max(comp, np.abs(mi * gi).max()); neg = min(neg, mi.min())
    if m.eq_fun is not None:
        h, Jh = m.user_eq(z)
        mh = mu[k:k + h.size]; k += h.size
        rhs += Jh.T @ mh
        viol = max(viol, np.abs(h).max())
    if nbnd_rows is not None:
        for (idx, sign), mb in zip(nbnd_rows, mu[k:]):
            rhs[idx] += sign * mb
            neg = min(neg, mb)
    lam = np.linalg.lstsq(Jc[:, :nfree].T, -rhs[:nfree], rcond=None)[0]
    stat = np.abs(rhs[:nfree] + Jc[:, :nfree].T @ lam).max() / max(1.0, np.abs(g).max())
    return stat, viol, comp, neg


@pytest.mark.parametrize("name,kw,B,hard,iters", [("vanderpol", dict(ph=10, ch=5, Ts=0.1), 64, True, 200),
                                                   ("ugv", dict(ph=30, ch=30), 24, False, 150),
                                                   ("osc6", dict(ph=20, ch=10, Ts=0.1), 4, True, 200)])
def test_gpu_solution_satisfies_the_restated_kkt_conditions(name, kw, B, hard, iters):
    """Independent of where scipy's SLSQP stops: stationarity (with the kernel's own multipliers), complementarity,
    dual and primal feasibility of the oracle's restatement of the reference problem at the point the GPU returns."""
    import torch
    from libmpc_amd.nlmpc import NLMPC, NLParameters, VANDERPOL, UGV, OSCILLATORS6
    rng = np.random.default_rng(41)
    m = dict(vanderpol=lambda: ref.vanderpol(**kw), ugv=lambda: ref.ugv(**kw), osc6=lambda: ref.oscillators(N=6, **kw))[name]()
    c = NLMPC(dict(vanderpol=VANDERPOL, ugv=UGV, osc6=OSCILLATORS6)[name], kw["ph"], kw["ch"], kw.get("Ts", 0.1))
    c.setOptimizerParameters(NLParameters(maximum_iteration=iters, hard_constraints=int(hard)))
    X0 = np.zeros((B, m.nx)); X0[:, :2] = rng.uniform(-0.5, 0.5, size=(B, 2))
    if name == "osc6":
        X0 = rng.uniform(-0.1, 0.1, size=(B, m.nx)); X0[:, 0] += 1.0
    U0 = np.zeros((B, m.nu))
    r = c.optimizeBatch(torch.from_numpy(X0), torch.from_numpy(U0), multipliers=True)
    torch.cuda.synchronize()
    z = r["z"].cpu().numpy(); mu = r["multipliers"].cpu().numpy(); st = r["solver_status"].cpu().numpy()
    worst = np.zeros(5)
    checked = 0
    for b in range(B):
        if st[b] not in (3, 4):
            continue
        stat, viol, comp, neg = _kkt_report(m, z[b], X0[b], mu[b], hard)
        worst = np.maximum(worst, [stat, viol, comp, -neg, np.abs(mu[b]).max()])
        checked += 1
    print("KKT %s: stationarity %.2e  violation %.2e  complementarity %.2e  most negative multiplier %.2e  (%d of %d instances)"
          % (name, worst[0], worst[1], worst[2], -worst[3], checked, B))
    assert checked >= 0.95 * B
    # the solve stops on the step length (tol_step), so the residual left is |B p| of a step just under that: a few 1e-5 of the
    # gradient's scale on ugv (3.7e-5 measured), below 1e-5 elsewhere; forward-difference noise (~1.5e-8 |f| / step) is smaller
    assert worst[0] <= 1e-4, worst
    # complementarity: a product multiplier x constraint value, the value up to the solver's constraint tolerance (1e-8: an instance that
    # ends on a stalled line search is accepted with rows that far out), the multipliers up to a few 1e3 on ugv (the 1e3 cost weights)
    assert worst[1] <= 1e-7 and worst[2] <= max(1e-6, 1e-8 * worst[4]) and worst[3] <= 1e-9, worst


def test_active_set_matches_the_oracle_optimum():
    """bit-exact active-set indices: the user inequalities with a non-zero multiplier are the ones the oracle's optimum holds
    at their bound (Van der Pol: u_i <= 0.5 active on the first moves for starts that need full effort)"""
    import torch
    from libmpc_amd.nlmpc import NLMPC, NLParameters, VANDERPOL
    kw = dict(ph=10, ch=5, Ts=0.1)
    m = ref.vanderpol(**kw)
    c = NLMPC(VANDERPOL, 10, 5, 0.1)
    c.setOptimizerParameters(NLParameters(maximum_iteration=200))
    X0 = np.array([[0.0, 1.0], [-0.9, -0.9], [0.8, -0.5], [-0.6, 0.9], [0.1, 0.1], [-1.0, 0.3]])
    U0 = np.zeros((6, 1))
    r = c.optimizeBatch(torch.from_numpy(X0), torch.from_numpy(U0), multipliers=True)
    torch.cuda.synchronize()
    mu = r["multipliers"].cpu().numpy()
    some_active = 0
    for b in range(6):
        o = m.solve(X0[b], U0[b], max_iter=1000)
        if not o["success"]:
            continue
        g = m.user_ineq(o["z"])[0]
        # move blocking repeats the last block's constraint: rows 4..10 are copies, the kernel keeps one of them in the set
        act_ref = {min(int(k), 4) for k in np.nonzero(g > -1e-7)[0]}
        act_gpu = {min(int(k), 4) for k in np.nonzero(mu[b] > 0)[0]}
        assert act_gpu == act_ref, (b, act_gpu, act_ref)
        some_active += bool(act_ref)
    assert some_active >= 2


def test_transcription_with_mapping_scalings_matches_oracle():
    """NLMPC::setInputScale / setStateScale (NLMPC.hpp:108,123): Mapping.hpp:174-211,221-257 and what the reference does
    and does not propagate into the derivatives (Objective.hpp:107-144, Constraints.hpp:269-284, 515-517, 565-571, 607-608)"""
    import torch
    from libmpc_amd.nlmpc import NLMPCEvaluator, VANDERPOL, UGV
    for name, kw, su, ss in [("vanderpol", dict(ph=10, ch=5, Ts=0.1), [2.0], [0.5, 4.0]),
                             ("ugv", dict(ph=12, ch=4), [3.0, 0.25], [2.0, 0.5, 1.5, 8.0])]:
        m = ref.vanderpol(**kw) if name == "vanderpol" else ref.ugv(**kw)
        ev = NLMPCEvaluator(VANDERPOL if name == "vanderpol" else UGV, kw["ph"], kw["ch"], kw.get("Ts", 0.1))
        ev.setInputScale(su); ev.setStateScale(ss)
        m.set_scaling(su, ss)
        rng = np.random.default_rng(9)
        B = 5
        Z = rng.normal(size=(B, m.nz)); Z[:, -1] = rng.normal(scale=0.1, size=B); X0 = rng.normal(size=(B, m.nx))
        out = ev.evaluate(torch.from_numpy(Z), torch.from_numpy(X0))
        torch.cuda.synchronize()
        Jd = ev.dense_eq_jacobian(out["jeq"])
        o = {k: v.cpu().numpy() for k, v in out.items()}
        for b in range(B):
            m.x0 = X0[b]
            f0, g = m.objective(Z[b]); c, J = m.state_eq(Z[b]); gi, Ji = m.user_ineq(Z[b])
            assert abs(o["cost"][b] - f0) <= 1e-12 * max(1.0, abs(f0))
            np.testing.assert_allclose(o["ceq"][b], c, rtol=0, atol=1e-13 * max(1.0, np.abs(c).max()))
            np.testing.assert_allclose(o["cineq"][b], gi, rtol=0, atol=1e-13)
            np.testing.assert_allclose(o["grad"][b], g, rtol=1e-9, atol=64 * np.finfo(float).eps * max(1.0, abs(f0)) / ref.DV)
            np.testing.assert_allclose(Jd[b], J, rtol=1e-9, atol=1e-6)
            np.testing.assert_allclose(o["jineq"][b], Ji, rtol=1e-9, atol=1e-6)


def test_nlopt_stopping_tolerances_stop_early_with_their_codes():
    """NLParameters::relative_ftol / absolute_ftol / relative_xtol / absolute_xtol (NLOptimizer.hpp:135-138): disabled (-1) the
    solve runs to its own convergence test (XTOL_REACHED = 4); a loose ftol stops earlier with FTOL_REACHED = 3 at nearly the
    same cost; a loose xtol stops earlier with 4; all map to SUCCESS (NLOptimizer.hpp:729-750)"""
    import torch
    from libmpc_amd.nlmpc import NLMPC, NLParameters, VANDERPOL
    rng = np.random.default_rng(3)
    B = 64
    x0 = torch.from_numpy(rng.uniform(-1, 1, size=(B, 2))); u0 = torch.zeros(B, 1, dtype=torch.float64)
    res = {}
    for key, prm in [("none", {}), ("ftol_rel", dict(relative_ftol=1e-4)), ("ftol_abs", dict(absolute_ftol=1e-3)),
                     ("xtol_rel", dict(relative_xtol=1e-2)), ("xtol_abs", dict(absolute_xtol=1e-2))]:
        c = NLMPC(VANDERPOL, 10, 5, 0.1)
        c.setOptimizerParameters(NLParameters(maximum_iteration=200, **prm))
        r = c.optimizeBatch(x0, u0)
        torch.cuda.synchronize()
        res[key] = {k: r[k].cpu().numpy() for k in ("status", "solver_status", "iterations", "cost")}
    base = res["none"]
    assert (base["solver_status"] == 4).mean() > 0.9
    for key, code in [("ftol_rel", 3), ("ftol_abs", 3), ("xtol_rel", 4), ("xtol_abs", 4)]:
        r = res[key]
        assert (r["status"] == 0).mean() > 0.9, key
        assert np.isin(r["solver_status"], (code, 4)).mean() > 0.9, (key, np.unique(r["solver_status"]))
        if code == 3:
            assert (r["solver_status"] == 3).mean() > 0.5, (key, np.unique(r["solver_status"], return_counts=True))
        assert r["iterations"].mean() < base["iterations"].mean(), (key, r["iterations"].mean(), base["iterations"].mean())
        ok = (r["status"] == 0) & (base["status"] == 0)
        assert np.abs(r["cost"][ok] - base["cost"][ok]).max() <= 2e-2 * np.maximum(1.0, base["cost"][ok]).max(), key


def test_control_gather_single_rank_through_the_c_abi():
    """mpcx_comm_* / mpcx_allgather_u: a one-rank RCCL communicator gathers a block onto itself on the caller's stream"""
    import torch
    from libmpc_amd.distributed import ControlGather, allgather_controls
    g = ControlGather(0, 0, 1)
    a = torch.arange(4096 * 4, dtype=torch.float64, device="cuda").reshape(4096, 4)
    out = g.allgather(a)
    torch.cuda.synchronize()
    assert out.shape == (4096, 4) and torch.equal(out, a)
    out2 = allgather_controls(a, gather=g)
    torch.cuda.synchronize()
    assert torch.equal(out2, a)


def test_per_instance_model_parameters_give_each_instance_its_own_controller_s_result():
    """mpcx_nlmpc_batch.params: every instance its own constants of the built-in system (each UGV its own obstacles and
    preferred velocity).  An instance solved with its parameters in the batch is bit for bit the instance solved by a
    controller created with those parameters."""
    import torch
    from libmpc_amd.nlmpc import NLMPC, NLParameters, UGV
    rng = np.random.default_rng(12)
    base = np.array([0.7071067811865476, 0.7071067811865476, 2.0, 1.0, 0.3, 1.0, 1.0, 0.3, 0.1])
    sets = [base.copy() for _ in range(4)]
    sets[1][2:5] = [1.5, 0.6, 0.4]                   # obstacle 0 elsewhere, larger
    sets[2][:2] = [1.0, 0.0]                         # another preferred velocity
    sets[3][5:8] = [0.5, 1.4, 0.2]                   # obstacle 1 elsewhere
    B = 16
    which = rng.integers(0, 4, size=B)
    P = np.stack([sets[k] for k in which])
    X0 = np.zeros((B, 4)); X0[:, :2] = rng.uniform(-0.5, 0.5, size=(B, 2))
    U0 = np.zeros((B, 2))
    c = NLMPC(UGV, 30, 30, 0.1)
    c.setOptimizerParameters(NLParameters(maximum_iteration=150, hard_constraints=0))
    r = c.optimizeBatch(torch.from_numpy(X0), torch.from_numpy(U0), params=torch.from_numpy(P))
    torch.cuda.synchronize()
    differs = 0
    for k in range(4):
        ck = NLMPC(UGV, 30, 30, 0.1, params=sets[k])
        ck.setOptimizerParameters(NLParameters(maximum_iteration=150, hard_constraints=0))
        idx = np.nonzero(which == k)[0]
        if idx.size == 0:
            continue
        rk = ck.optimizeBatch(torch.from_numpy(X0[idx]), torch.from_numpy(U0[idx]))
        torch.cuda.synchronize()
        sel = torch.from_numpy(idx).cuda()
        assert torch.equal(r["cmd"][sel], rk["cmd"]) and torch.equal(r["cost"][sel], rk["cost"])
        assert torch.equal(r["solver_status"][sel], rk["solver_status"])
        if k:
            r0 = c.optimizeBatch(torch.from_numpy(X0[idx]), torch.from_numpy(U0[idx]))       # the controller's own parameters
            torch.cuda.synchronize()
            differs += int(not torch.equal(r0["cmd"], rk["cmd"]))
    assert differs >= 2                              # the parameters matter
    v = NLMPC(1, 10, 5, 0.1)                         # Van der Pol has no parameters: refused
    with pytest.raises(Exception):
        v.optimizeBatch(torch.zeros(2, 2, dtype=torch.float64), torch.zeros(2, 1, dtype=torch.float64), params=torch.zeros(2, 1, dtype=torch.float64))


@pytest.mark.parametrize("name,kw,hard,iters", [("vanderpol", dict(ph=10, ch=5, Ts=0.1), True, 200), ("ugv", dict(ph=12, ch=4), False, 150)])
def test_shards_equal_rows_of_the_unsharded_solve(name, kw, hard, iters):
    """multi-GPU readiness on one device: a batch solved as four shards (what four ranks would do) gives, bit for bit, the rows of
    the unsharded solve -- commands, costs, statuses, decision vectors"""
    import torch
    from libmpc_amd.nlmpc import NLMPC, NLParameters, VANDERPOL, UGV
    m = NLMPC(dict(vanderpol=VANDERPOL, ugv=UGV)[name], kw["ph"], kw["ch"], kw.get("Ts", 0.1))
    m.setOptimizerParameters(NLParameters(maximum_iteration=iters, hard_constraints=int(hard)))
    rng = np.random.default_rng(11)
    Bs, R = 96, 4
    X0 = rng.uniform(-0.5, 0.5, size=(Bs * R, m.nx)); U0 = np.zeros((Bs * R, m.nu))
    if name == "ugv":
        X0[:, 2:] = 0.0
    whole = m.optimizeBatch(torch.from_numpy(X0), torch.from_numpy(U0)); torch.cuda.synchronize()
    for rk in range(R):
        sl = slice(rk * Bs, (rk + 1) * Bs)
        part = m.optimizeBatch(torch.from_numpy(X0[sl]), torch.from_numpy(U0[sl])); torch.cuda.synchronize()
        for key in ("cmd", "cost", "status", "z"):
            assert torch.equal(part[key], whole[key][sl]), (rk, key)


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw,hard,iters", [("vanderpol", dict(ph=10, ch=5, Ts=0.1), True, 200), ("ugv", dict(ph=12, ch=4), False, 150)])
def test_lds_resident_blocks_agree_with_the_workspace_form(name, kw, hard, iters, monkeypatch):
    """Small built-in systems keep the dynamics blocks and the sweeps' operands in the wavefront's LDS slice (nlmpc_plan,
    NlmpcDev::lds_blocks); MPCX_DEBUG_LDS_BLOCKS=0 at set-up keeps them in the workspace as the larger systems do.  Same arithmetic, another
    place: the two forms agree bit for bit."""
    import torch
    from libmpc_amd.nlmpc import NLMPC, NLParameters, VANDERPOL, UGV
    rng = np.random.default_rng(5)
    B = 96
    nx = 4 if name == "ugv" else 2
    X0 = np.zeros((B, nx)); X0[:, :2] = rng.uniform(-0.5, 0.5, size=(B, 2))
    U0 = np.zeros((B, 2 if name == "ugv" else 1))
    out = []
    monkeypatch.setenv("MPCX_NLMPC_FORM", "wave")          # (both variants are variants of the one-wavefront form)
    for blocks in ("1", "0"):
        monkeypatch.setenv("MPCX_DEBUG_LDS_BLOCKS", blocks)
        c = NLMPC(dict(vanderpol=VANDERPOL, ugv=UGV)[name], kw["ph"], kw["ch"], kw.get("Ts", 0.1))
        c.setOptimizerParameters(NLParameters(maximum_iteration=iters, hard_constraints=int(hard)))
        r = c.optimizeBatch(torch.from_numpy(X0), torch.from_numpy(U0))
        torch.cuda.synchronize()
        out.append({k: r[k].cpu().numpy() for k in ("cmd", "cost", "status", "iterations", "z")})
    a, b = out
    assert (a["status"] == 0).all()
    assert np.array_equal(a["iterations"], b["iterations"]) and np.array_equal(a["status"], b["status"])
    assert np.array_equal(a["cmd"], b["cmd"]) and np.array_equal(a["cost"], b["cost"]) and np.array_equal(a["z"], b["z"])


@pytest.mark.gpu
def test_shape_sweep_never_trips_the_exec_guard():
    """tools/nlmpc_stress.py in the wavefront form (the one that carries the guard): built-in and run-time compiled hook models over a range
    of shapes; no instance may end with nlopt's FORCED_STOP code (-5), which that kernel reports only when a phase boundary found lanes missing from
    EXEC (nlmpc_engine.hpp `exec_full`), and nothing may come back non-finite"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MPCX_NLMPC_FORM="wave")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "nlmpc_stress.py")], env=env, capture_output=True, text=True, timeout=900)
    print(out.stdout[-3000:])
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "0 problem(s), 0 instance(s) with the FORCED_STOP code" in out.stdout
