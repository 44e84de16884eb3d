"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's set-up utilities.

discretization: include/mpc/Utils.hpp:23-47 (A, B), :63-89 (A, B, Be) -- zero-order hold through one matrix
exponential of [[A B]; [0 0]] * Ts (Eigen's MatrixFunctions::exp; here scipy.linalg.expm, the same Pade scaling-and-
squaring family).  Pinned by test/test_utils.cpp:10-63 (double integrator chain, Ts = 0.02) in tests/test_oracle.py.
"""
import numpy as np
from scipy.linalg import expm


def discretization(A, B, Ts, Be=None):
    A = np.asarray(A, float); B = np.asarray(B, float)
    nx, nu = B.shape
    blocks = [A, B] + ([] if Be is None else [np.asarray(Be, float)])
    top = np.hstack(blocks) * Ts
    n = top.shape[1]
    M = np.zeros((n, n)); M[:nx] = top
    E = expm(M)
    out = (E[:nx, :nx], E[:nx, nx:nx + nu])
    return out if Be is None else out + (E[:nx, nx + nu:],)
