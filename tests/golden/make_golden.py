"""Regenerates tests/golden/quadrotor_oracle_n20.npz from the CPU oracle.

The reference itself cannot be run here (Eigen3/OSQP/NLopt absent), so these
vectors are *derived*: oracle outputs on the synthetic batch of SURVEY.md 8(d), whose instance 0
is the reference's own pinned test input.  Inputs + expected outputs only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import quadrotor_oracle  # noqa: E402
from oracle.lmpc_numpy import quadrotor_batch  # noqa: E402

if __name__ == "__main__":
    B, ph = 64, 20
    x0, u0, yref = quadrotor_batch(B)
    o = quadrotor_oracle(ph)
    r = o.solve_batch_constref(x0, u0, yref, want_active=True)
    lo = [np.nonzero(r["active_lower"][b][o.neq:])[0] + o.neq for b in range(B)]
    up = [np.nonzero(r["active_upper"][b][o.neq:])[0] + o.neq for b in range(B)]
    w = max(max(len(a) for a in lo), max(len(a) for a in up), 1)
    pad = lambda rows: np.array([np.pad(a, (0, w - len(a)), constant_values=-1) for a in rows], dtype=np.int32)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "quadrotor_oracle_n20.npz"),
                        x0=x0, u0=u0, yref=yref, cmd=r["cmd"], cost=r["cost"], status=r["status"],
                        polished=r["polished"], active_lower=pad(lo), active_upper=pad(up))
    print("written", B, "instances; polished", int((r["polished"] == 1).sum()))
