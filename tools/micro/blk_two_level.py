"""DESIGN.md section 9-2, second finding: nlmpc_sqp with the dynamics blocks AND the defects c in LDS next to a two-level working-set factor
returned wrong results at config 3 in round 3; the plan never selects the combination.  On the probe build (make -C libmpc_amd/csrc probe:
nlmpc_sqp<Model, true, true> compiled in) MPCX_DEBUG_LDS_BLOCKS=2 forces it; this solves the golden UGV instances both ways and compares
them with each other and with the committed oracle solutions.  Usage: python tools/micro/blk_two_level.py [n_cases]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np
    import torch
    from libmpc_amd.nlmpc import NLMPC, NLParameters, UGV
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "nlmpc_oracle_solutions.json")))["ugv_ph30_ch30"]["cases"][:int(sys.argv[2])]
    c = NLMPC(UGV, 30, 30, 0.1)
    c.setOptimizerParameters(NLParameters(maximum_iteration=150, hard_constraints=0))
    x0 = np.array([k["x0"] for k in g]); u0 = np.array([k["u0"] for k in g])
    r = c.optimizeBatch(torch.from_numpy(x0), torch.from_numpy(u0))
    torch.cuda.synchronize()
    # config 3's batch as bench.py draws it: 4096 instances, every CU loaded
    import bench
    c2, X0, U0 = bench.nl_make("ugv", 4096)
    r2 = c2.optimizeBatch(torch.from_numpy(X0), torch.from_numpy(U0))
    torch.cuda.synchronize()
    print(json.dumps({"cmd": r["cmd"].cpu().numpy().tolist(), "status": r["status"].cpu().numpy().tolist(),
                      "solver_status": r["solver_status"].cpu().numpy().tolist(), "iterations": r["iterations"].cpu().numpy().tolist(),
                      "cost": r["cost"].cpu().numpy().tolist(),
                      "big_cmd": r2["cmd"].cpu().numpy().tolist(), "big_status": r2["status"].cpu().numpy().tolist(),
                      "big_iterations": r2["iterations"].cpu().numpy().tolist()}))
    sys.exit(0)

import numpy as np  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = json.load(open(os.path.join(ROOT, "tests", "golden", "nlmpc_oracle_solutions.json")))["ugv_ph30_ch30"]["cases"][:n]
res = {}
for tag, blocks in (("workspace blocks", "0"), ("LDS blocks + two-level factor", "2")):
    env = dict(os.environ, MPCX_LIBRARY=os.path.join(ROOT, "libmpc_amd", "libmpcx_probe.so"), MPCX_NLMPC_FORM="wave", MPCX_DEBUG_LDS_BLOCKS=blocks,
               MPCX_DEBUG_OCCUPANCY="1")
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n)], env=env, capture_output=True, text=True, timeout=600)
    occ = [ln for ln in out.stderr.splitlines() if ln.startswith("nlmpc_sqp")]
    print("%s: %s" % (tag, occ[0] if occ else "(no launch line) " + out.stderr[-400:]))
    res[tag] = json.loads(out.stdout.strip().splitlines()[-1])
a, b = res["workspace blocks"], res["LDS blocks + two-level factor"]
usable = [k["success"] or (k["slsqp_mode"] == 8 and k["eq_violation"] < 1e-8 and k["ineq_violation"] < 1e-6) for k in g]
for tag, r in res.items():
    d = [np.abs(np.array(r["cmd"][i]) - np.array(g[i]["cmd"])).max() / max(1.0, np.abs(np.array(g[i]["cmd"])).max()) for i in range(n)]
    close = sum(1 for i in range(n) if usable[i] and d[i] <= 1e-5)
    print("%s: %d of %d usable golden cases within 1e-5, %d converged, iterations mean %.1f" %
          (tag, close, sum(usable), sum(1 for s in r["status"] if s == 0), float(np.mean(r["iterations"]))))
dd = [np.abs(np.array(a["cmd"][i]) - np.array(b["cmd"][i])).max() for i in range(n)]
bd = np.abs(np.array(a["big_cmd"]) - np.array(b["big_cmd"])).max(axis=1)
print("config 3's batch of 4096: max |cmd difference| %.3e, instances that differ at all %d, different status %d, different iteration count %d; converged %d / %d" %
      (bd.max(), int((bd > 0).sum()), sum(1 for x, y in zip(a["big_status"], b["big_status"]) if x != y),
       sum(1 for x, y in zip(a["big_iterations"], b["big_iterations"]) if x != y), sum(1 for x in a["big_status"] if x == 0), sum(1 for x in b["big_status"] if x == 0)))
print("the two forms against each other: max |cmd difference| %.3e; instances with different status %d, different iteration count %d" %
      (max(dd), sum(1 for i in range(n) if a["status"][i] != b["status"][i]), sum(1 for i in range(n) if a["iterations"][i] != b["iterations"][i])))
