"""Measurement script: six oscillators with |u| <= ub in a chosen form -- per instance the restated problem's KKT residuals with the kernel's
multipliers, and for the worst ones the oracle's SLSQP result.  Usage: python tools/osc_bounds_kkt.py [B] [ub] (form by MPCX_NLMPC_* in the environment)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import nlmpc_c, nlmpc_numpy as ref  # noqa: E402
from libmpc_amd.nlmpc import NLMPC, NLParameters, OSCILLATORS6  # noqa: E402
from test_nlmpc_gpu import _kkt_report  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ub = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
N, ph, ch = 6, 20, 10
rng = np.random.default_rng(2025 + N)
X0 = rng.uniform(-0.5, 0.5, size=(B, 2 * N)); X0[:, 0] += 1.0
c = NLMPC(OSCILLATORS6, ph, ch, 0.1)
c.setOptimizerParameters(NLParameters(maximum_iteration=200))
c.setInputBounds([-ub] * N, [ub] * N, (0, ch))
r = c.optimizeBatch(torch.from_numpy(X0), torch.zeros(B, N, dtype=torch.float64), multipliers=True); torch.cuda.synchronize()
r = {k: v.cpu().numpy() for k, v in r.items() if k != "_keep"}
mk = ref.oscillators(N=N, ph=ph, ch=ch, Ts=0.1)
rows = [(ph * 2 * N + k, sg) for k in range(ch * N) for sg in (1.0, -1.0)]
res = np.array([_kkt_report(mk, r["z"][b], X0[b], r["multipliers"][b], True, nbnd_rows=rows) for b in range(B)])
w14 = np.array([int(c.debug_workspace(i)["scal"][14]) for i in range(B)])
att = 1 + ((w14 >> 1) & 1)
print("form", c._lib.mpcx_nlmpc_last_form(c._h), "codes", {int(k): int((r["solver_status"] == k).sum()) for k in np.unique(r["solver_status"])}, "second attempts", int((att == 2).sum()), "left the inverse form on the way", int(((w14 >> 2) & 1).sum()))
print("stationarity: max %.2e, > 1e-4: %d, > 1e-5: %d; violation max %.2e; complementarity max %.2e; most negative multiplier %.2e"
      % (res[:, 0].max(), (res[:, 0] > 1e-4).sum(), (res[:, 0] > 1e-5).sum(), res[:, 1].max(), res[:, 2].max(), res[:, 3].min()))
def _oracle(b):
    return nlmpc_c.make("osc6").solve(X0[b], np.zeros(N), max_iter=400, hard=True, lb_u=[-ub] * N, ub_u=[ub] * N)


import multiprocessing as mp  # noqa: E402
with mp.get_context("fork").Pool(max(1, (os.cpu_count() or 2) - 2)) as pool:
    orc = pool.map(_oracle, range(B), chunksize=4)
ok = np.array([o["success"] for o in orc])
dc = np.array([(r["cost"][b] - orc[b]["cost"]) / abs(orc[b]["cost"]) for b in range(B)])
du = np.array([np.abs(r["cmd"][b] - orc[b]["cmd"]).max() for b in range(B)])
print("against the oracle on all %d (%d converged there): cost rel diff max %.2e (> 1e-8: %d), |cmd - oracle| max %.2e (> 1e-5: %d)"
      % (B, ok.sum(), np.abs(dc[ok]).max(), (np.abs(dc[ok]) > 1e-8).sum(), du[ok].max(), (du[ok] > 1e-5).sum()))
bad = [b for b in range(B) if ok[b] and (abs(dc[b]) > 1e-8 or du[b] > 1e-5)]
for b in (bad + list(np.argsort(-res[:, 0])[:3]))[:12]:
    o = orc[b]
    zu = r["z"][b][ph * 2 * N:-1]
    print("instance %d: stationarity %.2e its %d attempts %d cost %.10f oracle %.10f (rel %.1e) |cmd - oracle| %.1e inputs on a bound %d (oracle %d) most negative multiplier %.2e"
          % (b, res[b, 0], r["iterations"][b], att[b], r["cost"][b], o["cost"], (r["cost"][b] - o["cost"]) / o["cost"], np.abs(r["cmd"][b] - o["cmd"]).max(),
             (np.abs(np.abs(zu) - ub) <= 1e-9).sum(), (np.abs(np.abs(o["z"][ph * 2 * N:-1]) - ub) <= 1e-9).sum(), res[b, 3]))
