"""-m gpu: the NLMPC transcription kernels (mpcx_nlmpc_evaluate_batch) against the oracle's restatement of
Mapping / Objective / Constraints (oracle/nlmpc_numpy.py), on the reference's example systems."""
import numpy as np
import pytest

from oracle import nlmpc_numpy as ref

pytestmark = pytest.mark.gpu


def _cases():
    return [("vanderpol", dict(ph=10, ch=5, Ts=0.1)), ("vanderpol", dict(ph=10, ch=10, Ts=0.1)),
            ("vanderpol", dict(ph=7, ch=3, Ts=0.05)), ("ugv", dict(ph=30, ch=30)), ("ugv", dict(ph=12, ch=4))]


def _make(name, kw):
    from libmpc_amd.nlmpc import NLMPCEvaluator, VANDERPOL, UGV
    if name == "vanderpol":
        return ref.vanderpol(**kw), NLMPCEvaluator(VANDERPOL, kw["ph"], kw["ch"], kw["Ts"])
    return ref.ugv(**kw), NLMPCEvaluator(UGV, kw["ph"], kw["ch"], 0.1)


@pytest.mark.parametrize("name,kw", _cases())
def test_transcription_matches_oracle(name, kw):
    import torch
    m, ev = _make(name, kw)
    assert (ev.nz, ev.neq, ev.nineq) == (m.nz, m.ph * m.nx, m.ineq)
    rng = np.random.default_rng(7)
    B = 9
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
