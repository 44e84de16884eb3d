"""Shape sweep of the SQP kernels (testing aid): every built-in system over a range of horizons / move blockings / bounds,
64 random instances each, and the run-time compiled (hipRTC) hook models of tests/test_nlmpc_hooks.py; reports the solved fraction and flags
anything that is not finite, and any instance that ended with nlopt's FORCED_STOP code (-5) -- the wavefront form reports that code in one case
only: a phase boundary found lanes missing from EXEC (nlmpc_engine.hpp, `exec_full`).  A crash shows up as such.
The kernel form follows MPCX_NLMPC_FORM (run it once with =wave and once with =wg)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from libmpc_amd.nlmpc import NLMPC, NLParameters, OSCILLATORS6, OSCILLATORS8, UGV, VANDERPOL, VANDERPOL_TERMINAL  # noqa: E402

rng = np.random.default_rng(7)
cases = []
for ph, ch in ((5, 2), (10, 5), (10, 10), (17, 3), (20, 20)):
    cases.append(("vanderpol", VANDERPOL, 2, 1, ph, ch, True))
cases.append(("vdp-terminal", VANDERPOL_TERMINAL, 2, 1, 10, 5, True))
for ph, ch in ((8, 8), (12, 4), (30, 30), (30, 10), (25, 25)):
    cases.append(("ugv", UGV, 4, 2, ph, ch, False))
for ph, ch in ((10, 5), (20, 10), (20, 20), (25, 8)):
    cases.append(("osc6", OSCILLATORS6, 12, 6, ph, ch, True))
for ph, ch in ((10, 5), (30, 15), (20, 20), (12, 12)):
    cases.append(("osc8", OSCILLATORS8, 16, 8, ph, ch, True))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_nlmpc_hooks as hooks  # noqa: E402  (the hook bodies as C++ text)
for ph, ch in ((10, 5), (7, 3)):
    cases.append(("vanderpol-rtc", ("rtc", 2, 1, 2, ph + 1, hooks.VDP), 2, 1, ph, ch, True))
for ph, ch in ((12, 4), (20, 20)):
    cases.append(("ugv-rtc", ("rtc", 4, 2, 4, 2 * (ph + 1), hooks.UGV), 4, 2, ph, ch, False))
bad = 0
guard = 0
for name, mid, nx, nu, ph, ch, hard in cases:
    for bounds in (False, True):
        if isinstance(mid, tuple):
            _, nx_, nu_, ny_, ineq_, src = mid
            c = NLMPC.from_sources(nx_, nu_, ny_, ph, ch, ineq_, 0, 0.1 if name.startswith("vanderpol") else 0.0, **src)
        else:
            c = NLMPC(mid, ph, ch, 0.1)
        c.setOptimizerParameters(NLParameters(maximum_iteration=150, hard_constraints=int(hard)))
        if bounds:
            c.setInputBounds(np.full(nu, -0.8), np.full(nu, 0.45), (0, ch))
            if name.startswith("vanderpol") or name.startswith("osc"):
                c.setStateBounds(np.full(nx, -3.0), np.full(nx, 3.0), (0, ph))
        B = 64
        x0 = rng.uniform(-0.3, 0.3, size=(B, nx))
        if name.startswith("osc"):
            x0[:, 0] += 1.0
        try:
            r = c.optimizeBatch(torch.from_numpy(x0), torch.zeros(B, nu, dtype=torch.float64), sequences=True)
        except Exception as e:                              # MPCX_NLMPC_FORM=wg forces the workgroup form: a shape its LDS plan does not take
            if os.environ.get("MPCX_NLMPC_FORM") == "wg" and "launch failed (-2)" in str(e):     # (many state bounds: dense rows beyond 160 KB)
                print("%-13s ph %2d ch %2d bounds %d: not taken by the workgroup form (the launcher's default falls back to nlmpc_sqp)" % (name, ph, ch, int(bounds)))
                continue
            raise
        torch.cuda.synchronize()
        st = r["status"].cpu().numpy(); cmd = r["cmd"].cpu().numpy()
        ok = st != 3
        finite = np.isfinite(cmd).all() and np.isfinite(r["seq_state"].cpu().numpy()[ok]).all()
        bad += int(not finite)
        guard += int((r["solver_status"].cpu().numpy() == -5).sum())
        print("%-13s ph %2d ch %2d bounds %d: converged %5.1f %%  iterations %5.1f  %s" %
              (name, ph, ch, int(bounds), 100.0 * ok.mean(), r["iterations"].float().mean().item(), "" if finite else "NOT FINITE"))
print("shape sweep done, %d problem(s), %d instance(s) with the FORCED_STOP code (EXEC guard)" % (bad, guard))
sys.exit(1 if bad or guard else 0)
