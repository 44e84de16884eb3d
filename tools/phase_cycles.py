"""Profiling aid: per-phase shader-cycle breakdown of lmpc_solve_kernel (median over instances)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc

ph = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
c = quadrotor_lmpc(ph, device=0)
x0, u0, yref = quadrotor_batch(B)
buf = torch.zeros((B, 8), dtype=torch.int64, device="cuda")
c._lib.mpcx_lmpc_debug_set_cycle_buffer(c._h, C.c_void_p(buf.data_ptr()))
batch, res, keep = c.make_batch(x0, u0, yref=yref)
for _ in range(3):
    c.launch(batch)
torch.cuda.synchronize()
t = buf.cpu().numpy()
d = np.diff(t[:, :4], axis=1)
names = ["load workspace", "solve(polish/admm)", "unpack"]
print("per-instance wave cycles: median / p90 / max (s_memtime ticks, 100 MHz const clock if readcyclecounter maps to s_memrealtime)")
for k, n in enumerate(names):
    print(f"  {n:22s} {np.median(d[:, k]):10.0f} {np.percentile(d[:, k], 90):10.0f} {d[:, k].max():10.0f}")
tot = t[:, 3] - t[:, 0]
print("  total                  %10.0f %10.0f %10.0f" % (np.median(tot), np.percentile(tot, 90), tot.max()))
print("span first start -> last end:", t[:, 3].max() - t[:, 0].min())
it = res.iterations.cpu().numpy()
print("iterations hist:", np.unique(it, return_counts=True))
ms = c.time_launches(batch, 20)
print("both kernels ms", ms)
ms2 = (C.c_float * 3)()
c._lib.mpcx_lmpc_debug_time_kernels(c._h, C.byref(batch), C.c_void_p(torch.cuda.current_stream().cuda_stream), 20, ms2)
print("assemble ms %.4f  polish-solve ms %.4f  admm-fallback ms %.4f" % (ms2[0], ms2[1], ms2[2]))
rd = res.polish_rounds.cpu().numpy()
na = res.active_count.cpu().numpy()
A = np.stack([np.ones_like(rd, dtype=float), rd.astype(float)], axis=1)
coef, *_ = np.linalg.lstsq(A, d[:, 1].astype(float), rcond=None)
print("solve cycles ~ %.0f + %.0f x rounds (least squares over instances); rounds mean %.2f max %d" % (coef[0], coef[1], rd.mean(), rd.max()))
for lo, hi in ((0, 4), (5, 6), (7, 8), (9, 12), (13, 16)):
    sel = (na >= lo) & (na <= hi) & (rd >= 2)
    if sel.sum() > 4:
        print("   final |A| in [%d, %d]: %5d instances, cycles per round (median) %.0f" % (lo, hi, sel.sum(), np.median(d[sel, 1] / rd[sel])))
start = t[:, 0] - t[:, 0].min(); end = t[:, 3] - t[:, 0].min()
order = np.argsort(end)[::-1][:8]
print("last finishers: (instance, rounds, start, end, own cycles)")
for i in order:
    print("   ", int(i), int(rd[i]), int(start[i]), int(end[i]), int(tot[i]))
print("cycles per round (median over instances with >=3 rounds):", np.median((d[:, 1] / np.maximum(rd, 1))[rd >= 3]))
print("start-time distribution: p50 %d p90 %d max %d" % (np.median(start), np.percentile(start, 90), start.max()))
import collections
late = start > np.percentile(start, 75)
print("instances starting late (4th quartile): mean rounds %.2f ; early: %.2f" % (rd[late].mean(), rd[~late].mean()))
pa = t[:, 4:8].astype(np.float64)
if pa.sum() > 0:
    sel = rd >= 3
    per = pa[sel] / rd[sel, None]
    print("per-round cycles (median): ws-build+hash %.0f | gather+factor+solve %.0f | w = t0 - Y[:,A] lam %.0f | KKT check + repair %.0f" % tuple(np.median(per, axis=0)))
