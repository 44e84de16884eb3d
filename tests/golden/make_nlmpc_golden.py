"""Generates tests/golden/nlmpc_oracle_solutions.json: optimal first moves / costs of the NLMPC oracle
(oracle/nlmpc_numpy.py: the reference's transcription restated + scipy SLSQP) on instances that take the oracle too long
to solve inside the test suite.  Run from the repository root: python tests/golden/make_nlmpc_golden.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nlmpc_numpy as ref  # noqa: E402

out = {}
# BASELINE config 5: eight coupled oscillators, ph = 30, ch = 15, x0 = e_0 and one perturbed start of its synthetic batch
m = ref.oscillators(N=8, ph=30, ch=15)
rng = np.random.default_rng(0)
x0s = [np.eye(16)[0], rng.uniform(-0.1, 0.1, size=16) + np.eye(16)[0]]
cases = []
for x0 in x0s:
    o = m.solve(x0, np.zeros(8), max_iter=200)
    cases.append(dict(x0=x0.tolist(), u0=[0.0] * 8, cmd=o["cmd"].tolist(), cost=o["cost"], success=bool(o["success"]), nit=o["nit"]))
out["oscillators8_ph30_ch15"] = dict(model="oscillators", N=8, ph=30, ch=15, Ts=0.1, hard=True, cases=cases)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "nlmpc_oracle_solutions.json"), "w"), indent=1)
print(json.dumps(out)[:400])
