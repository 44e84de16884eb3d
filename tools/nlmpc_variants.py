"""NLMPC: the forms of the solve kernel side by side in one process (the launcher reads MPCX_NLMPC_FORM / _WAVES / _BLOCKS per call):
time of one batched solve, solves/s, iterations, solved fraction, the form the launcher took.
Usage: python tools/nlmpc_variants.py [spec ...]   spec = workload:batch:FORM[:WAVES[:BLOCKS]]  (FORM: default | wg | wave; '-' = unset)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.nlmpc_bench import make  # noqa: E402

DEFAULT = ["ugv:4096:default", "ugv:4096:wg:4:0", "ugv:4096:wg:4:1", "ugv:4096:wave", "ugv:256:wg:4:1", "ugv:512:wg:4:1", "ugv:768:wg:4:0",
           "osc8:1024:default", "osc8:1024:wg:8", "osc8:1024:wg:4", "osc8:1024:wave", "osc8:256:wg:8", "osc8:256:wg:4",
           "osc6:1024:default", "osc6:1024:wg:4", "osc6:1024:wg:8", "osc6:1024:wave", "vanderpol:4096:default"]
for spec in (sys.argv[1:] or DEFAULT):
    f = spec.split(":")
    name, B, form = f[0], int(f[1]), f[2]
    for k in ("MPCX_NLMPC_FORM", "MPCX_NLMPC_WAVES", "MPCX_NLMPC_BLOCKS"):
        os.environ.pop(k, None)
    if form != "default":
        os.environ["MPCX_NLMPC_FORM"] = form
    if len(f) > 3 and f[3] != "-":
        os.environ["MPCX_NLMPC_WAVES"] = f[3]
    if len(f) > 4 and f[4] != "-":
        os.environ["MPCX_NLMPC_BLOCKS"] = f[4]
    try:
        c, x0, u0 = make(name, B)
        b, out = c.make_batch(x0, u0)
        c.time_launches(b, 1)
        reps = 2 if name != "vanderpol" else 20
        ms = c.time_launches(b, reps)
        torch.cuda.synchronize()
        st = out["solver_status"].cpu().numpy(); it = out["iterations"].cpu().numpy()
        got = int(c._lib.mpcx_nlmpc_last_form(c._h))
        print("%-28s form %d  %9.3f ms  %10.1f solves/s  iterations %.1f (max %d)  solved %.5f" %
              (spec, got, ms, B / ms * 1e3, it.mean(), it.max(), float((st > 0).mean())), flush=True)
    except Exception as e:  # a variant the plan refuses
        print("%-28s refused: %s" % (spec, str(e)[:120]), flush=True)
