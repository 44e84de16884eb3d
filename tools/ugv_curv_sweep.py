"""Measurement script: config 3 (UGV, non-convex obstacle rows) with the curvature estimate set to the condensed Gauss-Newton Hessian before
iteration K (MPCX_NLMPC_CURV0_IT=K; K = 0: from the start; MPCX_NLMPC_CURV0=0: never, the identity as NLopt's SLSQP) -- how many of the golden
oracle solutions are reached, how many instances end at another local optimum, and the throughput at the quoted batch.  One process per K."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import numpy as np
    import torch
    from libmpc_amd.nlmpc import NLMPC, NLParameters, UGV
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "nlmpc_oracle_solutions.json")))["ugv_ph30_ch30"]
    c = NLMPC(UGV, 30, 30, 0.1)
    c.setOptimizerParameters(NLParameters(maximum_iteration=150, hard_constraints=0))
    X0 = np.array([k["x0"] for k in gold["cases"]]); U0 = np.array([k["u0"] for k in gold["cases"]])
    r = c.optimizeBatch(torch.from_numpy(X0), torch.from_numpy(U0)); torch.cuda.synchronize()
    cmd, cost, st = r["cmd"].cpu().numpy(), r["cost"].cpu().numpy(), r["status"].cpu().numpy()
    usable = [k["success"] or (k["slsqp_mode"] == 8 and k["eq_violation"] < 1e-8 and k["ineq_violation"] < 1e-6) for k in gold["cases"]]
    agree = other_worse = other_better = failed = 0
    for b, k in enumerate(gold["cases"]):
        if not usable[b]:
            continue
        if st[b] == 3:
            failed += 1
        elif np.allclose(cmd[b], k["cmd"], rtol=1e-5, atol=1e-5):
            agree += 1
        elif cost[b] > k["cost"] * (1 + 2e-6):
            other_worse += 1
        else:
            other_better += 1
    rng = np.random.default_rng(0)
    B = 4096
    Xb = np.zeros((B, 4)); Xb[:, :2] = rng.uniform(-0.5, 0.5, size=(B, 2))
    b, rr = c.make_batch(torch.from_numpy(Xb), torch.zeros(B, 2, dtype=torch.float64))
    import ctypes as C
    from libmpc_amd._capi import check
    check(c._lib.mpcx_nlmpc_solve_batch(c._h, C.byref(b), torch.cuda.current_stream().cuda_stream)); torch.cuda.synchronize()
    ms = c.time_launches(b, 3)
    # the same 4096 instances from the identity (the route NLopt's SLSQP takes): how many end at the same command, how many at a better / worse cost
    os.environ["MPCX_NLMPC_CURV0"] = "0"
    ci = NLMPC(UGV, 30, 30, 0.1)
    ci.setOptimizerParameters(NLParameters(maximum_iteration=150, hard_constraints=0))
    ri = ci.optimizeBatch(torch.from_numpy(Xb), torch.zeros(B, 2, dtype=torch.float64)); torch.cuda.synchronize()
    ca, cb = rr["cmd"].cpu().numpy(), ri["cmd"].cpu().numpy()
    fa, fb = rr["cost"].cpu().numpy(), ri["cost"].cpu().numpy()
    both = (rr["status"].cpu().numpy() == 0) & (ri["status"].cpu().numpy() == 0)
    same = np.all(np.abs(ca - cb) <= 1e-5 * np.maximum(1.0, np.abs(cb)), axis=1)
    batch = dict(same=int((same & both).sum()), both_solved=int(both.sum()), worse=int((~same & both & (fa > fb * (1 + 2e-6))).sum()), better=int((~same & both & (fa < fb * (1 - 2e-6))).sum()))
    print("RESULT " + json.dumps(dict(agree=agree, usable=sum(usable), worse=other_worse, better_or_equal=other_better, failed=failed, ms=ms, solves_per_s=B / ms * 1e3,
                                      iters=float(rr["iterations"].double().mean()), solved=float((rr["status"] == 0).double().mean()), batch_vs_identity=batch)))


if __name__ == "__main__":
    if "--child" in sys.argv:
        child(); sys.exit(0)
    for env in [dict(MPCX_NLMPC_CURV0="0")] + [dict(MPCX_NLMPC_CURV0_IT=str(k)) for k in (sys.argv[1:] or ["0", "5", "10", "20", "30", "40"])]:
        e = dict(os.environ); e.update(env)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=e, capture_output=True, text=True)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
        print(env, line[0][7:] if line else p.stderr[-1500:], flush=True)
