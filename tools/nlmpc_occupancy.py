"""NLMPC: what a CU delivers against the workgroups (instances) it holds.  Every instance of the batch is the SAME problem (no spread of
iteration counts, no tail), the batch k x 256: each of the 256 CUs gets k workgroups; prints the time of the batched solve and the
instances per second of one CU.  Usage: python tools/nlmpc_occupancy.py [ugv|osc6|osc8|vanderpol]  (forms through MPCX_NLMPC_FORM / _WAVES /
_BLOCKS as everywhere)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from libmpc_amd._capi import check  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "ugv"
ks = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4, 6, 8, 16]
for k in ks:
    B = 256 * k
    c, x0, u0 = bench.nl_make(name, 8, device=0)
    x0 = np.repeat(x0[3:4], B, axis=0); u0 = np.repeat(u0[3:4], B, axis=0)
    b, out = c.make_batch(torch.from_numpy(x0), torch.from_numpy(u0))
    s = torch.cuda.current_stream(0).cuda_stream
    check(c._lib.mpcx_nlmpc_solve_batch(c._h, C.byref(b), s)); torch.cuda.synchronize()
    ms = c.time_launches(b, 2, s); torch.cuda.synchronize()
    it = out["iterations"].cpu().numpy()
    form = int(c._lib.mpcx_nlmpc_last_form(c._h))
    print(f"{name} k={k:2d} per CU (batch {B:5d}) form {form}: {ms:8.3f} ms, {it[0]} iterations, {k / ms:7.4f} instances per ms and CU, "
          f"{B / ms:8.1f} solves/ms", flush=True)
