"""Batched evaluation of libmpc++'s NLMPC transcription on the MI355X.

`NLMPCEvaluator` stands where the four NLopt callback trampolines of the reference's
`NLOptimizer<>` stand (include/mpc/NLMPC/NLOptimizer.hpp:760-997): given decision vectors it
returns the cost and its gradient, the dynamics equalities with their Jacobian blocks and the user
inequalities with theirs -- for a whole batch in one kernel launch through the C ABI
(`mpcx_nlmpc_*`, include/mpcx.h).  The system/cost/constraint hooks are the library's built-in
device functors (the reference's example systems); arbitrary host callables cannot run in a kernel.
No CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._capi import check

VANDERPOL, UGV = 1, 2


class NLMPCEvaluator:
    def __init__(self, model, ph, ch, Ts, params=None, device=0):
        self._lib = _capi.lib()
        self._h = C.c_void_p()
        self.device = device
        prm = None if params is None else np.ascontiguousarray(params, dtype=np.float64)
        check(self._lib.mpcx_nlmpc_create(int(model), int(ph), int(ch), float(Ts),
                                          None if prm is None else prm.ctypes.data, 0 if prm is None else prm.size,
                                          int(device), C.byref(self._h)))
        d = _capi.NlmpcDims()
        check(self._lib.mpcx_nlmpc_get_dims(self._h, C.byref(d)))
        self.nx, self.nu, self.ph, self.ch, self.nz, self.neq, self.nineq, self.jeq_w = (
            d.nx, d.nu, d.ph, d.ch, d.nz, d.neq, d.nineq, d.jeq_w)

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.mpcx_nlmpc_destroy(self._h)
            self._h = None

    def evaluate(self, z, x0, *, cost=True, grad=True, eq=True, eq_jac=True, ineq=True, ineq_jac=True, stream=None):
        """z [B, nz], x0 [B, nx]: fp64 tensors on the evaluator's device.  Returns a dict of device tensors."""
        import torch
        dev = torch.device("cuda", self.device)
        z = z.to(dev, torch.float64).contiguous(); x0 = x0.to(dev, torch.float64).contiguous()
        B = z.shape[0]
        assert z.shape == (B, self.nz) and x0.shape == (B, self.nx)
        mk = lambda on, *shape: torch.empty((B,) + shape, dtype=torch.float64, device=dev) if on else None
        out = dict(cost=mk(cost), grad=mk(grad, self.nz), ceq=mk(eq, self.neq), jeq=mk(eq_jac, self.ph, self.nx, self.jeq_w),
                   cineq=mk(ineq, self.nineq), jineq=mk(ineq_jac, self.nineq, self.nz))
        ptr = lambda t: None if t is None else t.data_ptr()
        s = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        check(self._lib.mpcx_nlmpc_evaluate_batch(self._h, B, z.data_ptr(), x0.data_ptr(), ptr(out["cost"]), ptr(out["grad"]),
                                                  ptr(out["ceq"]), ptr(out["jeq"]), ptr(out["cineq"]), ptr(out["jineq"]), s))
        return out

    def dense_eq_jacobian(self, jeq):
        """Scatter the Jacobian blocks into the reference's dense [neq x nz] layout (Constraints.hpp:455-482)."""
        nx, nu, ph, ch, nz = self.nx, self.nu, self.ph, self.ch, self.nz
        jb = jeq.detach().cpu().numpy()
        J = np.zeros((jb.shape[0], ph * nx, nz))
        for i in range(ph):
            r = slice(i * nx, (i + 1) * nx)
            if i > 0:
                J[:, r, (i - 1) * nx:i * nx] = jb[:, i, :, :nx]
            J[:, r, i * nx:(i + 1) * nx] = jb[:, i, :, nx:2 * nx]
            b = min(i, ch - 1)
            J[:, r, ph * nx + b * nu: ph * nx + (b + 1) * nu] += jb[:, i, :, 2 * nx:]
        return J
