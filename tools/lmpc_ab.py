"""A/B on the GPU box: the LMPC solve as two kernels (assemble -> workspace -> solve), with the record computed inside the solve kernel
by one mat-vec (persistent with the composed map in LDS from 1024 instances on), and as one workgroup per sixteen instances
(MFMA assemble into LDS, then one wavefront per instance).
Prints ms per step, per-kernel times and the largest difference of the results to the first variant."""
import ctypes as C
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch
from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ph = int(sys.argv[2]) if len(sys.argv) > 2 else 20
x0, u0, yref = quadrotor_batch(B)
ref = None
for name, fused in (("two-kernel", 0), ("fused-matvec", 1), ("group", 2), ("two-kernel", 0), ("group", 2)):
    c = quadrotor_lmpc(ph, device=0)
    c.debug_use_fused(fused)
    b, r, keep = c.make_batch(x0, u0, yref=yref)
    s = torch.cuda.current_stream(0)
    for _ in range(30):
        c.launch(b, s)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(300):
        c.launch(b, s)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 300 * 1e3
    ms3 = (C.c_float * 3)()
    c._lib.mpcx_lmpc_debug_time_kernels(c._h, C.byref(b), C.c_void_p(s.cuda_stream), 100, ms3)
    c.launch(b, s); torch.cuda.synchronize()
    out = (r.cmd.cpu().numpy().copy(), r.cost.cpu().numpy().copy(), r.status.cpu().numpy().copy())
    if ref is None:
        ref = out
    dcmd = np.abs(out[0] - ref[0]).max(); dcost = np.abs(out[1] - ref[1]).max() / np.abs(ref[1]).max()
    print("%-11s batch %d N %d ms/step %.4f" % (name, B, ph, dt), "kernels", [round(v, 4) for v in ms3], "rounds mean %.2f max %d" %
          (r.polish_rounds.float().mean().item(), r.polish_rounds.max().item()), "admm iters max %d" % r.iterations.max().item(),
          "solved", (r.status == 0).float().mean().item(),
          "max|dcmd| %.2e rel dcost %.2e status equal %s" % (dcmd, dcost, np.array_equal(out[2], ref[2])), flush=True)
