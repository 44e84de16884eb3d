"""CPU tests: the NLMPC transcription restatement (oracle/nlmpc_numpy.py) against the reference's
own component known answers: unwrap layout, mapping dimensions, objective value,
collocation residual and Jacobian, user-constraint passthrough.  The end-to-end NLMPC solve has no
reference known answer (parity unpinned at the NLopt boundary); the solve tests below check KKT-type
properties only."""
import numpy as np
import pytest

from oracle.nlmpc_numpy import DV, NlmpcRef, ugv, vanderpol


@pytest.mark.parametrize("ch", [1, 4, 7])
def test_unwrap_layout(ch):
    """test/NLMPC/test_common.cpp:46-106"""
    nx, nu, ph = 5, 3, 7
    m = NlmpcRef(nx, nu, 1, ph, ch, 1, 1)
    z = np.arange(m.nz, dtype=float)
    m.x0 = -np.arange(1, nx + 1, dtype=float)
    X, U, e = m.unwrap(z)
    assert np.array_equal(X[0], m.x0)
    for i in range(1, ph + 1):
        assert np.array_equal(X[i], z[(i - 1) * nx:i * nx])
    u_index = 0
    for i in range(ph + 1):
        if i < ch:
            u_index = ph * nx + i * nu
        assert np.array_equal(U[i], z[u_index:u_index + nu])
    assert e == z[-1]


def test_mapping_dimensions():
    """test/NLMPC/test_common.cpp:9-44"""
    m = NlmpcRef(5, 3, 1, 7, 4, 1, 1)
    assert m.Iz2u.shape == (7 * 3, 4 * 3) and m.Iu2z.shape == (4 * 3, 7 * 3)
    # first ch-1 moves last one step, the last one ph-ch+1 steps
    assert np.array_equal(m.Iz2u.sum(axis=0), np.array([1] * 9 + [4] * 3, float))


def test_objective_value():
    """test/NLMPC/test_objective.cpp:47-62: sum of squares of X and U (last input row counted twice)"""
    m = NlmpcRef(5, 3, 1, 7, 7, 0, 0)
    m.cost = lambda X, Y, U, e: np.sum(X * X) + np.sum(U * U)
    m.x0 = np.zeros(5)
    v, g = m.objective(np.arange(m.nz, dtype=float), want_grad=True)
    assert v == 65730.0
    # gradient: forward differences with the reference's step quirk; analytic value 2 z (4 z on the paired row)
    z = np.arange(m.nz, dtype=float)
    exact = 2 * z; exact[-4:-1] *= 2; exact[-1] = 0
    assert np.allclose(g, exact, rtol=1e-5, atol=1e-5)


def test_collocation_residual_and_jacobian():
    """test/NLMPC/test_constraints.cpp:60-142 (van der Pol, Ts = 0.01, z = 0..6, x0 = 0)"""
    m = NlmpcRef(2, 1, 1, 2, 2, 0, 0)
    m.continuous = True; m.Ts = 0.01
    m.f = lambda x, u, p: np.array([(1.0 - x[1] * x[1]) * x[0] - x[1] + u[0], x[0]])
    m.x0 = np.zeros(2)
    z = np.arange(m.nz, dtype=float)
    c, J0 = m.state_eq(z, want_jac=False)
    assert np.allclose(c, [0.035, -1, -2.05, -1.99], atol=1e-3) and not J0.any()
    c, J = m.state_eq(z, want_jac=True)
    Jexp = np.array([[-1, -0.005, 0, 0, 0.01, 0, 0], [0.005, -1, 0, 0, 0, 0, 0],
                     [1, -0.005, -1.04, -0.065, 0, 0.01, 0], [0.005, 1, 0.005, -1, 0, 0, 0]])
    assert np.allclose(J, Jexp, atol=1e-3)


def test_user_inequality_passthrough():
    """test/NLMPC/test_constraints.cpp:144-213: g = x(0,0) -> value x0[0], zero Jacobian"""
    m = NlmpcRef(2, 1, 1, 5, 5, 1, 0)
    m.continuous = True
    m.f = lambda x, u, p: np.array([(1.0 - x[1] * x[1]) * x[0] - x[1] + u[0], x[0]])
    m.ineq_fun = lambda X, Y, U, e: np.array([X[0, 0]])
    m.x0 = np.array([10.0, 0.0])
    g, J = m.user_ineq(np.arange(m.nz, dtype=float))
    assert g[0] == 10.0 and not J.any()


def test_fd_step_quirk_is_reproduced():
    """the whole-horizon finite differences take their step from element (row j, column 0), not from the
    perturbed element (Objective.hpp:217, Constraints.hpp:661,688)"""
    m = NlmpcRef(2, 1, 1, 3, 3, 0, 0)
    seen = []
    def cost(X, Y, U, e):
        seen.append(X.copy())
        return 0.0
    m.cost = cost
    m.x0 = np.array([5.0, 0.0])
    z = np.zeros(m.nz); z[2] = 100.0        # X[2,0] = 100: its own |x| would give a step of 100*dv
    m.objective(z, want_grad=True)
    base = seen[0]
    # perturbation j=0 uses Xa(0) = |X[0,0]| = 5 for every horizon step; j=1 uses Xa(1) = |X[1,0]| -> max(0,1) = 1
    assert np.isclose(seen[1][1, 0] - base[1, 0], DV * 5.0)
    assert np.isclose(seen[3][2, 0] - base[2, 0], DV * 5.0)
    assert np.isclose(seen[2][1, 1] - base[1, 1], DV * 1.0)


def test_vanderpol_solve_properties():
    """config 1: examples/vanderpol_ex.cpp first solve from x = (0, 1).  No reference known answer
    exists; check feasibility of the transcription and the input bound."""
    m = vanderpol()
    r = m.solve([0.0, 1.0], [0.0], max_iter=200)
    assert r["success"]
    c, _ = m.state_eq(r["z"], want_jac=False)
    assert np.abs(c).max() < 1e-8
    assert (r["U"][:, 0] <= 0.5 + 1e-8).all()
    assert 0 < r["cost"] < 20


def test_ugv_functions_shapes():
    m = ugv(ph=10, ch=10)
    m.x0 = np.zeros(4)
    z = np.zeros(m.nz)
    v, g = m.objective(z)
    gi, J = m.user_ineq(z)
    c, Je = m.state_eq(z)
    assert g.shape == (m.nz,) and gi.shape == (22,) and J.shape == (22, m.nz) and Je.shape == (40, m.nz)
    assert np.isclose(v, 1e3 * 11 * 1.0)       # |0 - v_pref|^2 = 1 at each of the 11 steps


def test_oscillators_model_shapes_and_field():
    from oracle.nlmpc_numpy import oscillators
    m = oscillators(N=6, ph=20, ch=10)
    assert (m.nz, m.ph * m.nx, m.ineq) == (301, 240, 126)          # osc-ref sizes
    x = np.zeros(12); x[0] = 1.0
    dx = m.f(x, np.zeros(6), 0)
    assert abs(dx[1] + 1.5) < 1e-15 and abs(dx[3] - 0.1) < 1e-15 and dx[0] == 0.0   # -x0 - 5 k x0 on itself, k x0 on a neighbour


def test_user_equality_constraints_known_answer():
    """test/NLMPC/test_constraints.cpp:211-274: eq_con[0] = x(0, 0) with x0 = (10, 0) -> value 10, Jacobian zero"""
    m = NlmpcRef(2, 1, 1, 5, 5, 0, eq=1)
    m.continuous = True; m.Ts = 0.1
    m.f = lambda x, u, p: np.array([(1.0 - x[1] * x[1]) * x[0] - x[1] + u[0], x[0]])
    m.eq_fun = lambda X, U: np.array([X[0, 0]])
    m.x0 = np.array([10.0, 0.0])
    z = np.arange(m.nz, dtype=float)
    h, J = m.user_eq(z)
    assert h.tolist() == [10.0] and not J.any()
    # a constraint that does depend on the decision variables: the Jacobian is the derivative, the last input row pairs
    m.eq_fun = lambda X, U: np.array([X[5, 1] ** 2 + 3.0 * U[4, 0] + U[5, 0]])
    h, J = m.user_eq(z)
    assert abs(h[0] - (z[9] ** 2 + 4.0 * z[14])) < 1e-12
    expect = np.zeros(m.nz); expect[9] = 2 * z[9]; expect[14] = 4.0
    np.testing.assert_allclose(J[0], expect, rtol=1e-6, atol=1e-6)


def test_terminal_constraint_solve_property():
    from oracle.nlmpc_numpy import vanderpol_terminal
    m = vanderpol_terminal(ph=10, ch=10)
    o = m.solve([0.1, 0.1], [0.0], max_iter=500)
    assert o["success"] and np.abs(o["X"][10]).max() < 1e-8 and (o["U"][:, 0] <= 0.5 + 1e-9).all()


@pytest.mark.parametrize("name", ["vanderpol", "ugv", "osc6", "osc8"])
def test_compiled_callbacks_equal_the_numpy_restatement(name):
    """oracle/nlmpc_callbacks.c (bench.py's compiled NLMPC CPU baseline) against oracle/nlmpc_numpy.py (pinned above by the
    reference's component known answers): cost and constraint values to round-off, the central-difference Jacobians to their
    noise, the forward-difference gradient to its (one ulp of the cost over a 1.5e-8 step)"""
    from oracle import nlmpc_c
    from oracle import nlmpc_numpy as N
    a = dict(vanderpol=lambda: N.vanderpol(10, 5, 0.1), ugv=lambda: N.ugv(30, 30), osc6=lambda: N.oscillators(6, 20, 10),
             osc8=lambda: N.oscillators(8, 30, 15))[name]()
    b = nlmpc_c.make(name)
    assert (a.nz, a.ineq) == (b.nz, b.nineq)
    rng = np.random.default_rng(5)
    for _ in range(3):
        x0 = rng.uniform(-0.5, 0.5, size=a.nx); a.x0 = x0; b.x0 = x0
        z = 0.7 * rng.normal(size=a.nz)
        fa, ga = a.objective(z); fb, gb = b.objective(z)
        assert abs(fa - fb) <= 1e-13 * abs(fa)
        assert np.abs(ga - gb).max() <= 2e-5 * max(1.0, np.abs(ga).max())
        ca, Ja = a.state_eq(z); cb, Jb = b.state_eq(z)
        assert np.abs(ca - cb).max() <= 1e-13 and np.abs(Ja - Jb).max() <= 1e-7
        ia, Ia = a.user_ineq(z); ib, Ib = b.user_ineq(z)
        assert np.abs(ia - ib).max() <= 1e-13 and np.abs(Ia - Ib).max() <= 1e-7


def test_compiled_baseline_solves_like_the_numpy_oracle():
    from oracle import nlmpc_c
    from oracle import nlmpc_numpy as N
    a, b = N.vanderpol(10, 5, 0.1), nlmpc_c.make("vanderpol")
    ra = a.solve([0.0, 1.0], [0.0], max_iter=200); rb = b.solve([0.0, 1.0], [0.0], max_iter=200)
    assert rb["success"] and abs(ra["cmd"][0] - rb["cmd"][0]) <= 1e-6 and abs(ra["cost"] - rb["cost"]) <= 1e-8 * abs(ra["cost"])
