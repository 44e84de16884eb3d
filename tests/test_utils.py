"""Set-up utility: zero-order-hold discretisation (reference include/mpc/Utils.hpp:23-89)."""
import numpy as np
import pytest

from oracle.utils_numpy import discretization as ref_c2d
from libmpc_amd.workloads import quadrotor_matrices


def _double_integrator_chain(dof=6):
    A = np.zeros((2 * dof, 2 * dof)); A[:dof, dof:] = np.eye(dof)
    B = np.zeros((2 * dof, dof)); B[dof:] = np.eye(dof)
    return A, B


def test_oracle_matches_reference_known_answer():
    """test/test_utils.cpp:10-63: chain of 6 double integrators at Ts = 0.02 -> Ad = [I 0.02 I; 0 I], Bd = [2e-4 I; 0.02 I]"""
    A, B = _double_integrator_chain()
    Ad, Bd = ref_c2d(A, B, 0.02)
    Ad_t = np.eye(12); Ad_t[:6, 6:] = 0.02 * np.eye(6)
    Bd_t = np.vstack([0.0002 * np.eye(6), 0.02 * np.eye(6)])
    assert np.allclose(Ad, Ad_t, rtol=1e-12, atol=1e-15) and np.allclose(Bd, Bd_t, rtol=1e-12, atol=1e-15)
    # with a disturbance matrix (Utils.hpp:63-89) the extra block is discretised like B
    Ad2, Bd2, Bed = ref_c2d(A, B, 0.02, Be=B[:, :2])
    assert np.allclose(Ad2, Ad) and np.allclose(Bd2, Bd) and np.allclose(Bed, Bd[:, :2])


@pytest.mark.gpu
def test_device_discretization_matches_oracle():
    import torch
    from libmpc_amd.utils import discretization
    A, B = _double_integrator_chain()
    Ad, Bd = discretization(A, B, 0.02)
    Ad_t = np.eye(12); Ad_t[:6, 6:] = 0.02 * np.eye(6)
    assert np.allclose(Ad[0].cpu().numpy(), Ad_t, rtol=1e-13, atol=1e-16)
    assert np.allclose(Bd[0].cpu().numpy(), np.vstack([0.0002 * np.eye(6), 0.02 * np.eye(6)]), rtol=1e-13, atol=1e-16)
    # a batch of random systems (stable and unstable, stiff and slow) with per-instance sampling times
    rng = np.random.default_rng(4)
    n, nx, nu = 300, 9, 4
    As = rng.normal(size=(n, nx, nx)) * rng.uniform(0.1, 30.0, size=(n, 1, 1))
    Bs = rng.normal(size=(n, nx, nu))
    Ts = rng.uniform(0.005, 0.3, size=n)
    Ad, Bd = discretization(torch.from_numpy(As), torch.from_numpy(Bs), torch.from_numpy(Ts))
    Ad, Bd = Ad.cpu().numpy(), Bd.cpu().numpy()
    for i in range(n):
        ra, rb = ref_c2d(As[i], Bs[i], Ts[i])
        sa = max(1.0, np.abs(ra).max())
        assert np.abs(Ad[i] - ra).max() <= 1e-11 * sa, (i, np.abs(Ad[i] - ra).max(), sa)
        assert np.abs(Bd[i] - rb).max() <= 1e-11 * max(1.0, np.abs(rb).max())


@pytest.mark.gpu
def test_discretized_model_feeds_the_controller():
    """heterogeneous set-up path: a continuous-time model discretised on the device gives the controller the same
    answer as the host-discretised one (semigroup property: exp(M Ts) = exp(M Ts/2)^2 checked on the way)"""
    from libmpc_amd.utils import discretization
    Ad, Bd, _ = quadrotor_matrices()
    # a continuous-time generator whose exponential is the example's discrete model does not exist in closed form; use
    # the semigroup property on a random generator instead
    rng = np.random.default_rng(9)
    A = rng.normal(size=(12, 12)); B = rng.normal(size=(12, 4))
    a1, b1 = discretization(A, B, 0.1)
    a2, b2 = discretization(A, B, 0.05)
    a1, b1, a2, b2 = (t[0].cpu().numpy() for t in (a1, b1, a2, b2))
    assert np.allclose(a1, a2 @ a2, rtol=1e-11, atol=1e-12)
    assert np.allclose(b1, a2 @ b2 + b2, rtol=1e-11, atol=1e-12)
