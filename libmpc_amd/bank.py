"""A mixed batch over a few different controllers.

The engine's batch is B instances of ONE controller (one model, one set of weights and bounds: what a `mpc::LMPC<>` object
is in the reference).  A fleet usually has a handful of vehicle variants, not B: `LMPCBank` holds K controllers, takes a
batch whose instances carry a controller index, solves each group on its own HIP stream (the handles are independent: own
workspace, own dispatch queues, so the groups overlap on the GPU) and returns the results in the caller's order.

This is heterogeneity by grouping -- no new kernel, every instance is solved by exactly the code path and with exactly the
results of its own controller.  `LMPCHetero` is the other end: K controllers (K up to the batch size: every instance its own
A, B, C, weights, bounds) behind ONE set of kernels, each instance reading its own model's factors from HBM
(mpcx_lmpc_hetero_*, include/mpcx.h).  torch is used for what it is here for: device memory, index_select / index_copy and
streams."""
from __future__ import annotations

import numpy as np
import torch

import ctypes as C

from . import _capi
from ._capi import check
from .lmpc import LMPC, BatchResult


def group_by_model(model, n_models: int):
    """Stable grouping of instance indices by controller index.  Returns (order, offsets): `order[offsets[k]:offsets[k+1]]`
    are the instances of controller k in their original relative order."""
    model = np.asarray(model, dtype=np.int64).reshape(-1)
    if model.size and (model.min() < 0 or model.max() >= n_models):
        raise ValueError("controller index out of range")
    order = np.argsort(model, kind="stable")
    counts = np.bincount(model, minlength=n_models)
    offsets = np.concatenate([[0], np.cumsum(counts)])
    return order, offsets


class LMPCBank:
    """K `libmpc_amd.LMPC` controllers of equal dimensions behind one `optimizeBatch(x0, lastU, model, ...)`."""

    def __init__(self, controllers):
        if not controllers:
            raise ValueError("at least one controller")
        c0 = controllers[0]
        for c in controllers:
            if (c.nx, c.nu, c.ny, c.ph, c.device) != (c0.nx, c0.nu, c0.ny, c0.ph, c0.device):
                raise ValueError("the controllers of a bank share their dimensions and their device")
        self.controllers = list(controllers)
        self.nx, self.nu, self.ny, self.ph, self.device = c0.nx, c0.nu, c0.ny, c0.ph, c0.device
        self._streams = [torch.cuda.Stream(device=self.device) for _ in controllers]

    def optimizeBatch(self, x0, lastU, model, yref=None, want_sequence=False) -> BatchResult:
        """x0 [B, nx], lastU [B, nu], model [B] (controller index per instance), yref None | [B, ny] | [B, ph, ny]."""
        dev = torch.device("cuda", self.device)
        x0 = torch.as_tensor(x0, dtype=torch.float64, device=dev)
        lastU = torch.as_tensor(lastU, dtype=torch.float64, device=dev)
        if yref is not None:
            yref = torch.as_tensor(yref, dtype=torch.float64, device=dev)
        B = x0.shape[0]
        order, offsets = group_by_model(model.cpu().numpy() if torch.is_tensor(model) else model, len(self.controllers))
        if order.size != B:
            raise ValueError("one controller index per instance")
        order_t = torch.from_numpy(order).to(dev)
        out = BatchResult(cmd=torch.empty((B, self.nu), dtype=torch.float64, device=dev),
                          cost=torch.empty(B, dtype=torch.float64, device=dev),
                          status=torch.empty(B, dtype=torch.int32, device=dev),
                          solver_status=torch.empty(B, dtype=torch.int32, device=dev),
                          is_feasible=torch.empty(B, dtype=torch.int32, device=dev),
                          iterations=torch.empty(B, dtype=torch.int32, device=dev))
        if want_sequence:
            out.seq_state = torch.empty((B, self.ph + 1, self.nx), dtype=torch.float64, device=dev)
            out.seq_output = torch.empty((B, self.ph + 1, self.ny), dtype=torch.float64, device=dev)
            out.seq_input = torch.empty((B, self.ph + 1, self.nu), dtype=torch.float64, device=dev)
        cur = torch.cuda.current_stream(dev)
        keep = []
        for k, c in enumerate(self.controllers):
            lo, hi = int(offsets[k]), int(offsets[k + 1])
            if hi == lo:
                continue
            s = self._streams[k]
            s.wait_stream(cur)                                   # the inputs were produced on the caller's stream
            with torch.cuda.stream(s):
                idx = order_t[lo:hi]
                xk, uk = x0.index_select(0, idx), lastU.index_select(0, idx)
                yk = None if yref is None else yref.index_select(0, idx)
                r = c.optimizeBatch(xk, uk, yref=yk, want_sequence=want_sequence, stream=s)
                for name in ("cmd", "cost", "status", "solver_status", "is_feasible", "iterations") + \
                        (("seq_state", "seq_output", "seq_input") if want_sequence else ()):
                    getattr(out, name).index_copy_(0, idx, getattr(r, name))
                keep.append((r, xk, uk, yk, idx))
            cur.wait_stream(s)                                   # the caller's stream sees the scattered results
        out._inputs = keep
        return out


class LMPCHetero:
    """K configured `libmpc_amd.LMPC` controllers (host-only handles, `device=-1`, are enough) of equal dimensions and equal
    pattern of finite bounds, solved together by one set of kernels: instance b uses controller `model[b]` (default: b).
    In the reference each of them is its own `mpc::LMPC<>` object (LMPC.hpp:751)."""

    def __init__(self, controllers, device=0, condense_on_host=False):
        """condense_on_host: every controller's prediction matrices, Hessian and factors on the host cores instead of by the
        device kernel (lmpc_hetero.hip; the default wherever the dimensions fit it)"""
        if not controllers:
            raise ValueError("at least one controller")
        c0 = controllers[0]
        self.nx, self.nu, self.ny, self.ndu, self.ph = c0.nx, c0.nu, c0.ny, c0.ndu, c0.ph
        self.device = int(device)
        self._lib = _capi.lib()
        arr = (C.c_void_p * len(controllers))(*[c._h for c in controllers])
        self._h = C.c_void_p()
        check(self._lib.mpcx_lmpc_hetero_create_ex(arr, len(controllers), self.device, int(bool(condense_on_host)), C.byref(self._h)))
        n, aw, mref, bpm = C.c_int(), C.c_int(), C.c_int(), C.c_double()
        check(self._lib.mpcx_lmpc_hetero_get_info(self._h, C.byref(n), C.byref(aw), C.byref(mref), C.byref(bpm)))
        self.count, self.active_words, self.m_ref, self.bytes_per_model = n.value, aw.value, mref.value, bpm.value
        self._template = c0

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.mpcx_lmpc_hetero_destroy(self._h)
            self._h = None

    def info(self):
        return {"active_words": self.active_words, "m_ref": self.m_ref}

    def debug_get(self, k, name):
        """testing aid: an O(n^3) array of controller k as the bank holds it on the device"""
        f = self._lib.mpcx_lmpc_hetero_debug_get
        n = check(f(self._h, int(k), name.encode(), None, 0))
        out = np.zeros(n)
        check(f(self._h, int(k), name.encode(), out.ctypes.data_as(C.c_void_p), n))
        return out

    # the descriptor is the single-controller one: borrow its builder (it only needs the dimensions and info())
    _torch, _dev, _ref = LMPC._torch, LMPC._dev, LMPC._ref

    def make_batch(self, x0, u0, model=None, **kw):
        b, res, keep = LMPC.make_batch(self, x0, u0, **kw)
        mi = None
        if model is not None:
            mi = torch.as_tensor(model).to(device=torch.device("cuda", self.device), dtype=torch.int32).contiguous()
            if mi.numel() != b.batch or (mi.numel() and (int(mi.min()) < 0 or int(mi.max()) >= self.count)):
                raise ValueError("model: one controller index in [0, %d) per instance" % self.count)
        return b, res, keep + (mi,), mi

    def launch(self, b, mi=None, stream=None):
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        check(self._lib.mpcx_lmpc_hetero_solve_batch(self._h, C.byref(b), None if mi is None else C.c_void_p(mi.data_ptr()), C.c_void_p(s.cuda_stream)))

    def time_launches(self, b, mi, repeats, stream=None):
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        ms = C.c_float()
        check(self._lib.mpcx_lmpc_hetero_time_solve_batch(self._h, C.byref(b), None if mi is None else C.c_void_p(mi.data_ptr()),
                                                          C.c_void_p(s.cuda_stream), int(repeats), C.byref(ms)))
        return ms.value

    def optimizeBatch(self, x0, lastU, model=None, yref=None, uref=None, duref=None, dmeas=None, want_active=False, want_sequence=False,
                      stream=None) -> BatchResult:
        """x0 [B, nx], lastU [B, nu]; model [B] controller index per instance (None: instance b = controller b); references None
        (each controller's own) | [B, n] | [B, ph, n]"""
        b, res, keep, mi = self.make_batch(x0, lastU, model, yref=yref, uref=uref, duref=duref, dmeas=dmeas, want_active=want_active,
                                           want_sequence=want_sequence)
        self.launch(b, mi, stream)
        res._inputs = keep
        return res
