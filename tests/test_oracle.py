"""CPU tests: the oracle against the reference's own known answers.

The reference cannot be compiled here (Eigen3 / OSQP v0.6.3 / NLopt absent); its tests' inline
literals are transcribed in tests/golden/reference_known_answers.json and re-expressed below."""
import json
import os

import numpy as np
import pytest
import scipy.linalg as sla

from helpers import INF, OracleFrontEnd, configure_quadrotor, quadrotor_oracle
from oracle import lmpc_numpy as LN
from oracle import osqp_numpy as ON
from oracle.lmpc_oracle import OracleLMPC

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "reference_known_answers.json")))


def test_quadrotor_known_answer():
    """test/LMPC/test_common.cpp:89-237"""
    g = GOLD["quadrotor_lmpc_n10"]
    f = configure_quadrotor(OracleFrontEnd(12, 4, 4, 12, 10, 10), 10)
    f.setOptimizerParameters(maximum_iteration=g["maximum_iteration"])
    r = f.optimize(np.array(g["x0"], float), np.array(g["u0"], float))
    expect = np.array(g["cmd"])
    assert np.linalg.norm(r["cmd"] - expect) <= g["rtol"] * min(np.linalg.norm(r["cmd"]), np.linalg.norm(expect))
    assert r["status"] == 0 and r["solver_status"] == 1 and r["is_feasible"] and r["polished"] == 1


@pytest.mark.parametrize("ph", [10, 20, 50])
def test_derived_values(ph):
    """values derived by an independent restatement; sizes of the reference QP"""
    d = GOLD["derived_survey_values"]["n%d" % ph]
    f = configure_quadrotor(OracleFrontEnd(12, 4, 4, 12, ph, ph), ph)
    f.setOptimizerParameters(maximum_iteration=250)
    r = f.optimize(np.zeros(12), np.zeros(4))
    assert abs(r["cmd"][1] - d["cmd1"]) < 1e-9 and abs(r["cmd"][0] + 0.9916) < 1e-12
    assert abs(r["cost"] - d["obj"]) < 1e-8
    sz = GOLD["qp_sizes"][str(ph)]
    assert (f.o.nvar, f.o.ncon) == (sz["n"], sz["m"])
    P, q, A, l, u = f.o.get_problem(np.zeros(12), np.zeros(4), f.yRef, f.uRef, f.duRef, f.dMeas, dense=True)
    assert np.count_nonzero(np.triu(P)) == sz["nnz_triu_P"] and np.count_nonzero(A) == sz["nnz_A"]
    if "active_lower_ineq" in d:
        assert list(np.nonzero(r["active_lower"][f.o.neq:])[0]) == d["active_lower_ineq"]
        assert not r["active_upper"][f.o.neq:].any()


def test_linear_default_constraints():
    """test/LMPC/test_constraints.cpp:169-204"""
    g = GOLD["linear_default_constraints"]; dd = g["dims"]
    o = OracleLMPC(dd["nx"], dd["nu"], dd["ndu"], dd["ny"], dd["ph"], dd["ch"])
    nx, nu, ph = dd["nx"], dd["nu"], dd["ph"]
    z = lambda r: np.zeros((r, ph))
    q, l, u = o.get_problem(np.full(nx, g["x0_fill"]), np.full(nu, g["u0_fill"]), z(dd["ny"]), z(nu), z(nu), z(dd["ndu"]))
    na = nx + nu
    for v in (l, u):
        assert (v[:nx] == g["l_x0"]).all() and (v[nx:na] == g["l_u0"]).all() and (v[na:(ph + 1) * na] == 0).all()
    assert (l[(ph + 1) * na:] == -INF).all() and (u[(ph + 1) * na:] == INF).all()


def test_linear_constraints_placement():
    """test/LMPC/test_constraints.cpp:206-295"""
    g = GOLD["linear_constraints"]; dd = g["dims"]
    nx, nu, ny, ph = dd["nx"], dd["nu"], dd["ny"], dd["ph"]
    o = OracleLMPC(nx, nu, dd["ndu"], ny, ph, dd["ch"])
    o.set_state_bounds(np.full((nx, ph), g["x_bounds"][0]), np.full((nx, ph), g["x_bounds"][1]))
    o.set_input_bounds(np.full((nu, ph), g["u_bounds"][0]), np.full((nu, ph), g["u_bounds"][1]))
    o.set_output_bounds(np.full((ny, ph), g["y_bounds"][0]), np.full((ny, ph), g["y_bounds"][1]))
    x0 = np.full(nx, g["x0_fill"]); u0 = np.full(nu, g["u0_fill"])
    o.set_scalar(np.full(ph, g["s_bounds"][0]), np.full(ph, g["s_bounds"][1]), x0, u0)
    z = lambda r: np.zeros((r, ph))
    q, l, u = o.get_problem(x0, u0, z(ny), z(nu), z(nu), z(dd["ndu"]))
    na = nx + nu; neq = (ph + 1) * na
    exp_l = np.tile(np.r_[np.full(nx, -1.0), np.full(nu, -3.0)], ph + 1)
    assert np.allclose(l[neq:neq + neq], exp_l) and np.allclose(u[neq:neq + neq], -exp_l)
    ys = slice(2 * neq, 2 * neq + (ph + 1) * ny)
    assert (l[ys] == -2).all() and (u[ys] == 2).all()
    du = slice(2 * neq + (ph + 1) * ny, 2 * neq + (ph + 1) * ny + ph * nu)
    assert (l[du] == -INF).all() and (u[du] == INF).all()
    assert (l[-ph:] == -4).all() and (u[-ph:] == 4).all()


def test_scalar_constraint_property():
    """test/LMPC/test_constraints.cpp:95-167 (c2d as include/mpc/Utils.hpp:23-47: zero-order hold)"""
    g = GOLD["scalar_constraint_property"]; dd = g["dims"]
    nx, nu, ph = dd["nx"], dd["nu"], dd["ph"]
    Ac = np.array(g["A_continuous"], float); Bc = np.array(g["B_continuous"], float)
    Mx = sla.expm(np.block([[Ac, Bc], [np.zeros((nu, nx + nu))]]) * g["Ts"])
    Ad, Bd = Mx[:nx, :nx], Mx[:nx, nx:]
    f = OracleFrontEnd(nx, nu, 0, dd["ny"], ph, dd["ch"])
    f.setStateSpaceModel(Ad, Bd, np.eye(2))
    assert f.setObjectiveWeights(g["OutputW"], g["InputW"], g["DeltaInputW"], (-1, -1))
    sX = np.ones(nx); sU = np.ones(nu)
    assert f.setScalarConstraint(g["smin"], g["smax"], sX, sU, (-1, -1))
    f.setOptimizerParameters(maximum_iteration=g["maximum_iteration"])
    r = f.optimize(np.array(g["x0"]), np.array(g["u0"]))
    for i in range(ph):
        s = sU @ r["input"][i] + sX @ r["state"][i]
        assert s <= g["smax"] + g["tol_upper"] and s >= g["smin"] - g["tol_lower"]


def test_output_mapping():
    """test/LMPC/test_common.cpp:239-280: y = C x + Dd d on the sequence's first row"""
    r = np.random.default_rng(1)
    nx, nu, ndu, ny, ph = 3, 1, 7, 6, 2
    f = OracleFrontEnd(nx, nu, ndu, ny, ph, ph)
    C = r.normal(size=(ny, nx)); Dd = r.normal(size=(ny, ndu))
    f.setStateSpaceModel(0.5 * np.eye(nx), np.ones((nx, nu)), C)
    f.setDisturbances(np.zeros((nx, ndu)), Dd)
    f.setObjectiveWeights(np.ones(ny), np.ones(nu), np.zeros(nu), (-1, -1))
    d = r.normal(size=ndu); x = r.normal(size=nx)
    f.setExogenousInputs(d, (-1, -1))
    out = f.optimize(x, np.zeros(nu))
    assert np.allclose(out["output"][0], C @ x + Dd @ d, atol=1e-7)


def test_c_oracle_matches_numpy_restatement():
    """two independent restatements (C + sparse LDL', numpy + dense LU) of the same algorithm"""
    ph = 10
    b = LN.quadrotor_builder(ph)
    x0, u0, yr = LN.quadrotor_batch(4)
    o = quadrotor_oracle(ph)
    z4 = np.zeros((4, ph))
    for i in range(4):
        yref = np.tile(yr[i][:, None], (1, ph))
        q, l, u = b.get(x0[i], u0[i], yref, z4, z4, z4)
        P2, q2, A2, l2, u2 = o.get_problem(x0[i], u0[i], yref, z4, z4, z4, dense=True)
        assert np.array_equal(P2, b.P) and np.array_equal(A2, b.A)
        assert np.allclose(q2, q, atol=1e-14) and np.array_equal(l2, l) and np.array_equal(u2, u)
        rn = ON.solve(b.P, q, b.A, l, u, ON.Settings(max_iter=250))
        rc = o.solve(x0[i], u0[i], yref, z4, z4, z4)
        assert rn["status"] == rc["solver_status"] and rn["iters"] == rc["iters"]
        assert np.allclose(rn["x"], rc["z"], rtol=1e-7, atol=1e-9)
        # equality rows are classified by the sign of multipliers that may be round-off zeros
        ne = o.neq
        assert np.array_equal(rn["active_lower"][ne:], rc["active_lower"][ne:])
        assert np.array_equal(rn["active_upper"][ne:], rc["active_upper"][ne:])


def test_golden_fixture_is_current():
    g = np.load(os.path.join(HERE, "golden", "quadrotor_oracle_n20.npz"))
    o = quadrotor_oracle(20)
    n = 16
    r = o.solve_batch_constref(g["x0"][:n], g["u0"][:n], g["yref"][:n], want_active=True)
    assert np.allclose(r["cmd"], g["cmd"][:n], rtol=1e-9, atol=1e-12)
    assert np.allclose(r["cost"], g["cost"][:n], rtol=1e-9)
    assert abs(g["cmd"][0][1] - GOLD["derived_survey_values"]["n20"]["cmd1"]) < 1e-9


def test_infeasible_instance_semantics():
    """x0 outside its step-0 box makes the QP infeasible.  With the
    reference's true infinities OSQP's certificate test evaluates inf*0 = NaN and never fires:
    the solve runs out of iterations and returns its last iterate as MAX_ITER_REACHED (feasible
    = true, sic, LOptimizer.hpp:344).  With OSQP-style finite infinities it is PRIMAL_INFEASIBLE."""
    x0 = np.zeros(12); x0[0] = 1.0          # |roll| <= pi/6 violated at step 0
    f = configure_quadrotor(OracleFrontEnd(12, 4, 4, 12, 10, 10), 10)
    f.setOptimizerParameters(maximum_iteration=250)
    r = f.optimize(x0, np.zeros(4))
    assert r["solver_status"] == -2 and r["status"] == 1 and r["is_feasible"] and np.isfinite(r["cmd"]).all()
    f.setOptimizerParameters(maximum_iteration=250, nan_faithful=0)
    r = f.optimize(x0, np.zeros(4))
    assert r["status"] == 2 and not r["is_feasible"] and np.isnan(r["cmd"]).all()
