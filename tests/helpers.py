"""Shared test helpers: build the same controller on the oracle and on the product."""
import math

import numpy as np

from oracle.lmpc_oracle import OracleLMPC, default_params

INF = float("inf")


def quadrotor_oracle(ph, ch=None, maximum_iteration=250):
    """examples/quadrotor_ex.cpp:52-93 on the CPU oracle (per-index setters, as slices {0,ph} do)."""
    from oracle.lmpc_numpy import quadrotor_model
    ch = ph if ch is None else ch
    o = OracleLMPC(12, 4, 4, 12, ph, ch)
    Ad, Bd, Cd = quadrotor_model()
    o.set_model(Ad, Bd, Cd)
    ow = np.array([0, 0, 10, 10, 10, 10, 0, 0, 0, 5, 5, 5.0]); uw = np.full(4, 0.1); duw = np.zeros(4)
    xmin = np.full(12, -INF); xmax = np.full(12, INF)
    xmin[0] = xmin[1] = -math.pi / 6; xmax[0] = xmax[1] = math.pi / 6; xmin[5] = -1
    umin = np.full(4, 9.6 - 10.5916); umax = np.full(4, 13 - 10.5916)
    for i in range(ph):
        o.set_objective_idx(i, ow, uw, duw)
        o.set_state_bounds_idx(i, xmin, xmax)
    for i in range(ch):
        o.set_input_bounds_idx(i, umin, umax)
    o.params = default_params(maximum_iteration=maximum_iteration)
    return o


def bits_to_rows(words, m):
    """[B, W] int32 bitmap words -> list of sorted row-index arrays"""
    w = np.asarray(words).astype(np.uint32)
    out = []
    for b in range(w.shape[0]):
        bits = np.unpackbits(w[b].view(np.uint8), bitorder="little")[:m]
        out.append(np.nonzero(bits)[0])
    return out
