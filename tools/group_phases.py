"""Profiling aid: where a workgroup of lmpc_solve_group spends its time (cycle stamps of the assemble phase and of the solves)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc

ph = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
c = quadrotor_lmpc(ph, device=0)
c.debug_use_fused(2)
x0, u0, yref = quadrotor_batch(B)
buf = torch.zeros((B, 8), dtype=torch.int64, device="cuda")
c._lib.mpcx_lmpc_debug_set_cycle_buffer(c._h, C.c_void_p(buf.data_ptr()))
batch, res, keep = c.make_batch(x0, u0, yref=yref)
for _ in range(3):
    c.launch(batch)
torch.cuda.synchronize()
t = buf.cpu().numpy().astype(np.float64)
# the cycle counter (s_memtime) is per XCD: stamps compare within a workgroup, not across workgroups
print("N=%d batch %d (cycles over instances)" % (ph, B))
for name, v in (("kernel entry -> inputs staged", t[:, 7] - t[:, 4]), ("first product", t[:, 5] - t[:, 7]), ("second product + tails", t[:, 6] - t[:, 5]),
                ("solve: slice -> first working set", t[:, 1] - t[:, 0]), ("solve: rounds", t[:, 2] - t[:, 1]), ("solve: unpack", t[:, 3] - t[:, 2]),
                ("wavefront start -> end of its instance", t[:, 3] - t[:, 4])):
    print("  %-44s min %9.0f p10 %9.0f median %9.0f max %9.0f" % (name, v.min(), np.percentile(v, 10), np.median(v), v.max()))
ge = t[: B // 16 * 16, 3].reshape(-1, 16); gs = t[: B // 16 * 16, 4].reshape(-1, 16)
wd = ge.max(axis=1) - gs.min(axis=1)
print("  workgroup: first start -> last end            min %9.0f p10 %9.0f median %9.0f max %9.0f" % (wd.min(), np.percentile(wd, 10), np.median(wd), wd.max()))
# per workgroup: spread of the wavefronts' start stamps (how long a workgroup of sixteen takes to be fully launched)
g = t[: B // 16 * 16, 4].reshape(-1, 16)
print("  spread of the 16 start stamps in a workgroup: median %.0f max %.0f" % (np.median(g.max(axis=1) - g.min(axis=1)), (g.max(axis=1) - g.min(axis=1)).max()))
ms = c.time_launches(batch, 50)
print("step ms %.4f" % ms)
# the instances that set the launch time: rounds and final working-set size of the ten slowest solves
try:
    rr = res.polish_rounds.cpu().numpy(); ac = res.active_count.cpu().numpy()
    dur = t[:, 3] - t[:, 0]
    order = np.argsort(-dur)[:10]
    print("slowest solves (instance: cycles, rounds, final working rows): " + "; ".join("%d: %.0f, %d, %d" % (i, dur[i], rr[i], ac[i]) for i in order))
    for k in range(1, int(rr.max()) + 1):
        m = rr == k
        if m.any():
            print("  %d rounds: %5d instances, solve cycles median %7.0f max %7.0f" % (k, m.sum(), np.median(dur[m]), dur[m].max()))
except Exception as e:      # (a result object without these arrays)
    print("no per-instance round counts:", e)
