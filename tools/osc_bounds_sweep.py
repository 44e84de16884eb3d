"""Measurement script (not product, not a test): the oscillator controllers with tight input bounds -- the problems at which the
inverse form of the working set's Schur complement (WgPlan::minv) lost instances in round 5 -- under every combination of
MPCX_NLMPC_MINV / MPCX_NLMPC_CARRY / MPCX_NLMPC_REFORM (each combination in its own process: the switches are read once, when a plan is
made).  Prints one line per (system, bound, combination): solved count, the other solver codes, milliseconds per launch.

Usage: python tools/osc_bounds_sweep.py [instances] [--child]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(n):
    import numpy as np
    import torch
    from libmpc_amd.nlmpc import NLMPC, NLParameters, OSCILLATORS6, OSCILLATORS8
    out = []
    for name, model, N, ph, ch in (("osc6", OSCILLATORS6, 6, 20, 10), ("osc8", OSCILLATORS8, 8, 30, 15)):
        for ub in (0.05, 0.15, None):
            rng = np.random.default_rng(77 + N)
            c = NLMPC(model, ph, ch, 0.1)
            c.setOptimizerParameters(NLParameters(maximum_iteration=200))
            if ub is not None:
                assert c.setInputBounds([-ub] * N, [ub] * N, (0, ch))
            X0 = rng.uniform(-0.5, 0.5, size=(n, 2 * N)); X0[:, 0] += 1.0
            b, r = c.make_batch(torch.from_numpy(X0), torch.zeros(n, N, dtype=torch.float64))
            import ctypes as C
            from libmpc_amd._capi import check
            s = torch.cuda.current_stream().cuda_stream
            check(c._lib.mpcx_nlmpc_solve_batch(c._h, C.byref(b), s)); torch.cuda.synchronize()
            ms = c.time_launches(b, 2)
            st = r["solver_status"].cpu().numpy()
            out.append(dict(system=name, ub=ub, n=n, solved=int((st > 0).sum()), ms=ms,
                            codes={int(k): int((st == k).sum()) for k in np.unique(st)},
                            iters=float(r["iterations"].double().mean()),
                            cmd_sum=float(r["cmd"].double().abs().sum())))
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1024
    if "--child" in sys.argv:
        child(n)
        sys.exit(0)
    combos = [dict(MPCX_NLMPC_MINV="0"), dict(MPCX_NLMPC_MINV="1", MPCX_NLMPC_CARRY="0"), dict(MPCX_NLMPC_MINV="1", MPCX_NLMPC_CARRY="1"), dict()]
    for extra in os.environ.get("SWEEP_EXTRA", "").split(";"):
        if extra:
            combos.append(dict(kv.split("=") for kv in extra.split(",")))
    for env in combos:
        e = dict(os.environ); e.update(env)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), str(n), "--child"], env=e, capture_output=True, text=True)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
        if not line:
            print("combo %s failed:\n%s\n%s" % (env, p.stdout[-2000:], p.stderr[-2000:]))
            continue
        for rec in json.loads(line[0][7:]):
            print("%-40s %s |u|<=%s: solved %d / %d, codes %s, %.2f ms, mean iters %.1f" % (env or "default", rec["system"], rec["ub"], rec["solved"], rec["n"], rec["codes"], rec["ms"], rec["iters"]), flush=True)
