"""Benchmark / test workloads: the reference's quadrotor controller and the synthetic
per-instance inputs of SURVEY.md section 8(d).  Data only -- no solver code."""
from __future__ import annotations

import math

import numpy as np

from .lmpc import LMPC, LParameters, inf


def quadrotor_matrices():
    """Ad, Bd, Cd of reference examples/quadrotor_ex.cpp:19-48 (numeric data)."""
    Ad = np.eye(12)
    for (i, j, v) in [(0, 6, 0.1), (1, 7, 0.1), (2, 8, 0.1),
                      (3, 0, 0.0488), (3, 6, 0.0016), (3, 9, 0.0992),
                      (4, 1, -0.0488), (4, 7, -0.0016), (4, 10, 0.0992),
                      (5, 11, 0.0992),
                      (9, 0, 0.9734), (9, 6, 0.0488), (9, 9, 0.9846),
                      (10, 1, -0.9734), (10, 7, -0.0488), (10, 10, 0.9846),
                      (11, 11, 0.9846)]:
        Ad[i, j] = v
    Bd = np.array([
        [0, -0.0726, 0, 0.0726], [-0.0726, 0, 0.0726, 0], [-0.0152, 0.0152, -0.0152, 0.0152],
        [0, -0.0006, -0.0000, 0.0006], [0.0006, 0, -0.0006, 0], [0.0106, 0.0106, 0.0106, 0.0106],
        [0, -1.4512, 0, 1.4512], [-1.4512, 0, 1.4512, 0], [-0.3049, 0.3049, -0.3049, 0.3049],
        [0, -0.0236, 0, 0.0236], [0.0236, 0, -0.0236, 0], [0.2107, 0.2107, 0.2107, 0.2107]], dtype=float)
    return Ad, Bd, np.eye(12)


def quadrotor_lmpc(ph=20, ch=None, device=0, maximum_iteration=250) -> LMPC:
    """The controller of examples/quadrotor_ex.cpp:52-93 at horizon ph (=ch by default)."""
    ch = ph if ch is None else ch
    c = LMPC(12, 4, 4, 12, ph, ch, device=device)
    Ad, Bd, Cd = quadrotor_matrices()
    c.setStateSpaceModel(Ad, Bd, Cd)
    c.setObjectiveWeights([0, 0, 10, 10, 10, 10, 0, 0, 0, 5, 5, 5], [0.1] * 4, [0] * 4, (0, ph))
    xmin = [-math.pi / 6, -math.pi / 6, -inf, -inf, -inf, -1] + [-inf] * 6
    xmax = [math.pi / 6, math.pi / 6] + [inf] * 10
    u0 = 10.5916
    c.setStateBounds(xmin, xmax, (0, ph))
    c.setOutputBounds([-inf] * 12, [inf] * 12, (0, ph))
    c.setInputBounds([9.6 - u0] * 4, [13 - u0] * 4, (0, ch))
    yref = np.zeros(12); yref[2] = 1.0
    c.setReferences(yref, np.zeros(4), np.zeros(4), (0, ph))
    c.setOptimizerParameters(LParameters(maximum_iteration=maximum_iteration))
    return c


def quadrotor_batch(B, first=0):
    """SplitMix64 inputs (seed 0x6d70632b2b + instance index, doubles = (r>>11)*2^-53):
    x0[0,1]~U(-.2,.2), x0[2..5]~U(-.5,.5), x0[6..11]~U(-.3,.3), u0~U(-.5,.5)^4,
    yRef[2]~U(.5,1.5).  Instance 0 is the reference test's exact input
    (test/LMPC/test_common.cpp:224-228).  Vectorised: draw k of instance i is
    mix(seed + i + (k+1)*golden) because SplitMix64 advances its state by a constant."""
    idx = (np.arange(B, dtype=np.uint64) + np.uint64(first))
    with np.errstate(over="ignore"):
        st = (np.uint64(0x6d70632b2b) + idx)[:, None] + (np.arange(1, 18, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))[None, :]
        z = st
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    vals = (z >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    w = np.array([0.2, 0.2] + [0.5] * 4 + [0.3] * 6)
    x0 = -w[None, :] + 2 * w[None, :] * vals[:, :12]
    u0 = -0.5 + vals[:, 12:16]
    yref = np.zeros((B, 12)); yref[:, 2] = 0.5 + vals[:, 16]
    if first == 0 and B > 0:
        x0[0] = 0; u0[0] = 0; yref[0] = 0; yref[0, 2] = 1.0
    return np.ascontiguousarray(x0), np.ascontiguousarray(u0), yref


def quadrotor_variant(k, ph=20, ch=None, device=0, maximum_iteration=250, into=None):
    """Variant k of the quadrotor controller for heterogeneous batches: dynamics scaled as by another mass / inertia / arm length,
    own output and input weights, own input limits and attitude limits -- the pattern of finite bounds is that of the example, so
    that every variant has the same constraint rows.  Deterministic in k (SplitMix64).  `into`: configure that front-end object
    (tests: the oracle's) instead of creating an LMPC."""
    ch = ph if ch is None else ch
    with np.errstate(over="ignore"):
        z = np.uint64(0x9E3779B97F4A7C15) * (np.arange(1, 13, dtype=np.uint64) + np.uint64(16 * k + 1))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    r = (z >> np.uint64(11)).astype(np.float64) * 2.0 ** -53          # 12 uniforms in [0, 1)
    Ad, Bd, Cd = quadrotor_matrices()
    Bv = Bd * (0.85 + 0.3 * r[0])                                     # thrust-to-mass ratio
    Bv[[0, 1, 3, 4, 6, 7, 9, 10], :] *= (0.9 + 0.2 * r[1])           # arm length / inertia
    Av = Ad.copy()
    Av[9, 9] = Ad[9, 9] * (0.98 + 0.03 * r[2]); Av[10, 10] = Ad[10, 10] * (0.98 + 0.03 * r[2]); Av[11, 11] = Ad[11, 11] * (0.98 + 0.03 * r[3])
    c = into if into is not None else LMPC(12, 4, 4, 12, ph, ch, device=device)
    c.setStateSpaceModel(Av, Bv, Cd)
    wy = np.array([0, 0, 10, 10, 10, 10, 0, 0, 0, 5, 5, 5], dtype=float) * (0.7 + 0.6 * r[4])
    c.setObjectiveWeights(wy, [0.1 * (0.5 + r[5])] * 4, [0] * 4, (0, ph))
    ang = math.pi / 6 * (0.8 + 0.4 * r[6])
    xmin = [-ang, -ang, -inf, -inf, -inf, -1 - 0.5 * r[7]] + [-inf] * 6
    xmax = [ang, ang] + [inf] * 10
    u0 = 10.5916
    c.setStateBounds(xmin, xmax, (0, ph))
    c.setOutputBounds([-inf] * 12, [inf] * 12, (0, ph))
    c.setInputBounds([9.6 - u0 - 0.3 * r[8]] * 4, [13 - u0 + 0.5 * r[9]] * 4, (0, ch))
    yref = np.zeros(12); yref[2] = 1.0
    c.setReferences(yref, np.zeros(4), np.zeros(4), (0, ph))
    if into is None:
        c.setOptimizerParameters(LParameters(maximum_iteration=maximum_iteration))
    return c
