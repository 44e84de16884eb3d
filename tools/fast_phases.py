"""Profiling aid for the lean LMPC solve kernel: per-phase cycles of a round (needs the library built with -DMPCX_PROFILE_ROUNDS,
MPCX_LIBRARY=libmpc_amd/libmpcx_prof.so).  Usage: fast_phases.py <ph> <batch>"""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc

ph = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
c = quadrotor_lmpc(ph, device=0)
c.debug_use_fused(0)          # the two-kernel form: lmpc_solve alone (the one-workgroup form files its own stamps in the same buffer)
x0, u0, yref = quadrotor_batch(B)
buf = torch.zeros((B, 8), dtype=torch.int64, device="cuda")
c._lib.mpcx_lmpc_debug_set_cycle_buffer(c._h, C.c_void_p(buf.data_ptr()))
batch, res, keep = c.make_batch(x0, u0, yref=yref)
for _ in range(3):
    c.launch(batch)
torch.cuda.synchronize()
t = buf.cpu().numpy().astype(np.float64)
rd = res.polish_rounds.cpu().numpy(); na = res.active_count.cpu().numpy()
names = ["ws build", "loads answered", "elimination", "w update", "multipliers+violations", "repair"]
print("N=%d batch %d: rounds mean %.2f max %d; load record %.0f, solve %.0f cycles (median)" % (ph, B, rd.mean(), rd.max(), np.median(t[:, 0]), np.median(t[:, 1])))
for lo, hi in ((0, 4), (5, 6), (7, 8), (9, 12), (13, 16), (0, 16)):
    sel = (na >= lo) & (na <= hi) & (rd >= 2)
    if sel.sum() < 4:
        continue
    per = t[sel, 2:8] / rd[sel, None]
    print("  final |A| in [%2d,%2d] (%5d instances): per round " % (lo, hi, sel.sum()) +
          " | ".join("%s %.0f" % (n, v) for n, v in zip(names, np.median(per, axis=0))) + " | sum %.0f of %.0f" % (np.median(per.sum(axis=1)), np.median(t[sel, 1] / rd[sel])))
