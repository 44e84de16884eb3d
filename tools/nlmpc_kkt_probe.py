"""Per-instance KKT figures of the UGV test batch (tests/test_nlmpc_gpu.py::test_gpu_solution_satisfies_the_restated_kkt_conditions) for one
form of the kernel (MPCX_NLMPC_FORM / _WAVES / _BLOCKS in the environment): largest violation, complementarity, iterations."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nlmpc_numpy as ref  # noqa: E402
from libmpc_amd.nlmpc import NLMPC, NLParameters, UGV  # noqa: E402

B = 24
m = ref.ugv(ph=30, ch=30)
rng = np.random.default_rng(41)
X0 = np.zeros((B, 4)); X0[:, :2] = rng.uniform(-0.5, 0.5, size=(B, 2))
c = NLMPC(UGV, 30, 30, 0.1)
c.setOptimizerParameters(NLParameters(maximum_iteration=150, hard_constraints=0))
r = c.optimizeBatch(torch.from_numpy(X0), torch.zeros(B, 2, dtype=torch.float64), multipliers=True)
torch.cuda.synchronize()
z = r["z"].cpu().numpy(); mu = r["multipliers"].cpu().numpy(); st = r["solver_status"].cpu().numpy(); it = r["iterations"].cpu().numpy()
worst = np.zeros(2)
for b in range(B):
    m.x0 = X0[b]
    gi, _ = m.user_ineq(z[b]); cc, _ = m.state_eq(z[b])
    v = [max(np.abs(cc).max(), gi.max()), np.abs(mu[b][:gi.size] * gi).max()]
    worst = np.maximum(worst, v)
    if v[0] > 1e-11 or v[1] > 1e-7:
        k = int(np.argmax(np.abs(mu[b][:gi.size] * gi)))
        print("  instance %d: %d iterations, status %d, violation %.2e, complementarity %.2e (row %d: g %.3e mu %.3e), cost %.12g" % (b, it[b], st[b], v[0], v[1], k, gi[k], mu[b][k], r["cost"][b].item()))
print("form %d: violation %.2e complementarity %.2e, iterations %s" % (int(c._lib.mpcx_nlmpc_last_form(c._h)), worst[0], worst[1], it.tolist()))
