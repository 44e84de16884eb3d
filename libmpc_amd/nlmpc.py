"""Batched evaluation of libmpc++'s NLMPC transcription on the MI355X.

`NLMPCEvaluator` stands where the four NLopt callback trampolines of the reference's
`NLOptimizer<>` stand (include/mpc/NLMPC/NLOptimizer.hpp:760-997): given decision vectors it
returns the cost and its gradient, the dynamics equalities with their Jacobian blocks and the user
inequalities with theirs -- for a whole batch in one kernel launch through the C ABI
(`mpcx_nlmpc_*`, include/mpcx.h).

The system / cost / constraint hooks are device code: either one of the reference's example systems built
into the library (`model=VANDERPOL` ...), or **user hooks given as C++ text** -- the bodies of the lambdas one
would hand to `setStateSpaceFunction`, `setObjectiveFunction`, `setIneqConFunction`, `setEqConFunction`,
`setOutputFunction` in the reference (NLMPC.hpp:139-281) -- compiled at run time for gfx950 (hipRTC,
`mpcx_nlmpc_create_from_source`).  A Python callable cannot run inside a kernel, its C++ spelling can.
No CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._capi import check

VANDERPOL, UGV, OSCILLATORS6, OSCILLATORS8, VANDERPOL_TERMINAL, VANDERPOL_RATE = 1, 2, 3, 4, 5, 6


def NLParameters(**kw) -> _capi.NLParams:
    """mpc::NLParameters with the reference defaults (Types.hpp:99-144)."""
    p = _capi.NLParams()
    _capi.lib().mpcx_nlparams_default(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


class NLMPCEvaluator:
    def __init__(self, model, ph, ch, Ts, params=None, device=0):
        self._lib = _capi.lib()
        self._h = C.c_void_p()
        self.device = device
        prm = None if params is None else np.ascontiguousarray(params, dtype=np.float64)
        check(self._lib.mpcx_nlmpc_create(int(model), int(ph), int(ch), float(Ts),
                                          None if prm is None else prm.ctypes.data, 0 if prm is None else prm.size,
                                          int(device), C.byref(self._h)))
        self._read_dims()

    @classmethod
    def from_sources(cls, nx, nu, ny, ph, ch, ineq, eq, Ts, *, state_fn, objective_fn, ineq_fn=None, eq_fn=None, output_fn=None,
                     preamble=None, device=0):
        """A controller whose hooks are the C++ bodies of the reference's lambdas (NLMPC.hpp:139-281), compiled at run time.

        Parameter names inside the bodies: state_fn (dx, x, u, step); objective_fn (x, y, u, e) -> return the cost;
        ineq_fn (in_con, x, y, u, e); eq_fn (eq_con, x, u); output_fn (y, x, u, step).  Types are the reference's
        (mpc::cvec<n>, mpc::mat<ph+1, n>), and num_states, num_inputs, num_output, pred_hor, ctrl_hor, ineq_c, eq_c are in
        scope.  Ts > 0: state_fn is dx/dt (setDiscretizationSamplingTime(Ts)); Ts <= 0: it returns x(k+1)."""
        self = cls.__new__(cls)
        self._lib = _capi.lib()
        self._h = C.c_void_p()
        self.device = device
        enc = lambda t: None if t is None else t.encode()
        src = _capi.NlmpcSource(int(nx), int(nu), int(ny), int(ph), int(ch), int(ineq), int(eq), enc(preamble), enc(state_fn),
                                enc(objective_fn), enc(ineq_fn), enc(eq_fn), enc(output_fn))
        check(self._lib.mpcx_nlmpc_create_from_source(C.byref(src), float(Ts), int(device), C.byref(self._h)))
        self._read_dims()
        return self

    def _read_dims(self):
        d = _capi.NlmpcDims()
        check(self._lib.mpcx_nlmpc_get_dims(self._h, C.byref(d)))
        self.nx, self.nu, self.ph, self.ch, self.nz, self.neq, self.nineq, self.jeq_w, self.neq_user, self.ny = (
            d.nx, d.nu, d.ph, d.ch, d.nz, d.neq, d.nineq, d.jeq_w, d.neq_user, d.ny)
        self.n_params = d.n_params

    @property
    def nbnd(self):
        """finite state / input bounds: the rows after the user constraints in `multipliers`"""
        d = _capi.NlmpcDims()
        check(self._lib.mpcx_nlmpc_get_dims(self._h, C.byref(d)))
        return d.nbnd

    def setInputScale(self, scaling):
        """NLMPC::setInputScale (NLMPC.hpp:108): the hooks see U = scaling * z_u"""
        v = np.ascontiguousarray(scaling, dtype=np.float64).reshape(self.nu)
        check(self._lib.mpcx_nlmpc_set_input_scale(self._h, v.ctypes.data))

    def setStateScale(self, scaling):
        """NLMPC::setStateScale (NLMPC.hpp:123): the hooks see X = [x0; z_x] / scaling"""
        v = np.ascontiguousarray(scaling, dtype=np.float64).reshape(self.nx)
        check(self._lib.mpcx_nlmpc_set_state_scale(self._h, v.ctypes.data))

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
                   cineq=mk(ineq, self.nineq + self.neq_user), jineq=mk(ineq_jac, self.nineq + self.neq_user, self.nz))
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


class NLMPC(NLMPCEvaluator):
    """Batched counterpart of `mpc::NLMPC<>` (reference include/mpc/NLMPC.hpp).  The hooks the reference takes as
    closures (`setStateSpaceFunction`, `setObjectiveFunction`, `setIneqConFunction`, ..., NLMPC.hpp:139-280) are either
    those of a built-in `model` or C++ text given to `NLMPC.from_sources(...)`; `setOptimizerParameters`, the bound and
    scale setters and `optimize` keep their meaning, and `optimizeBatch` runs B instances of NLOptimizer::run
    (NLOptimizer.hpp:412-638) in one kernel launch."""

    def setOptimizerParameters(self, p):
        check(self._lib.mpcx_nlmpc_set_optimizer_parameters(self._h, C.byref(p)))
        self._warm = bool(p.enable_warm_start)

    def _bounds(self, fn, lo, hi, n, horizon, slice_):
        lo = np.asarray(lo, dtype=np.float64); hi = np.asarray(hi, dtype=np.float64)
        if lo.ndim == 2:                        # matrix form: one column per step (NLMPC.hpp:285-316)
            ok = True
            for i in range(horizon):
                ok &= self._bounds(fn, lo[:, i], hi[:, i], n, horizon, (i, i + 1))
            return ok
        lo = np.ascontiguousarray(lo.reshape(n)); hi = np.ascontiguousarray(hi.reshape(n))
        a, b = (-1, -1) if slice_ is None else (slice_.start, slice_.end) if hasattr(slice_, "start") else slice_
        return fn(self._h, lo.ctypes.data, hi.ctypes.data, int(a), int(b)) == 0

    def setStateBounds(self, lo, hi, slice_=None):
        """NLMPC::setStateBounds (NLMPC.hpp:285-299, 346-358): returns False on an invalid slice, like the reference"""
        return self._bounds(self._lib.mpcx_nlmpc_set_state_bounds_slice, lo, hi, self.nx, self.ph, slice_)

    def setInputBounds(self, lo, hi, slice_=None):
        return self._bounds(self._lib.mpcx_nlmpc_set_input_bounds_slice, lo, hi, self.nu, self.ch, slice_)

    def setOutputBounds(self, *_a, **_k):
        raise RuntimeError("Output constraints cannot be set for this type of MPC")        # NLMPC.hpp:318-325

    def _closures_are_fixed(self, *_a, **_k):
        raise RuntimeError("a Python callable cannot run inside the kernel: give the hook bodies as C++ text to "
                           "NLMPC.from_sources(...), or use a built-in model")
    setStateSpaceFunction = setObjectiveFunction = setIneqConFunction = setEqConFunction = setOutputFunction = _closures_are_fixed

    def make_batch(self, x0, u0, z_warm=None, sequences=False, warm_curvature=False, multipliers=False, params=None):
        import torch
        dev = torch.device("cuda", self.device)
        x0 = x0.to(dev, torch.float64).contiguous(); u0 = u0.to(dev, torch.float64).contiguous()
        B = x0.shape[0]
        assert x0.shape == (B, self.nx) and u0.shape == (B, self.nu)
        f = lambda *sh: torch.empty((B,) + sh, dtype=torch.float64, device=dev)
        i = lambda: torch.empty(B, dtype=torch.int32, device=dev)
        out = dict(cmd=f(self.nu), cost=f(), status=i(), solver_status=i(), is_feasible=i(), iterations=i(), z=f(self.nz))
        if sequences:
            out["seq_state"] = f(self.ph + 1, self.nx); out["seq_input"] = f(self.ph + 1, self.nu)
            out["seq_output"] = f(self.ph + 1, self.ny)
        if multipliers:
            out["multipliers"] = f(self.nineq + self.neq_user + self.nbnd)
        zw = None if z_warm is None else z_warm.to(dev, torch.float64).contiguous()
        b = _capi.NlmpcBatch(batch=B, x0=x0.data_ptr(), u0=u0.data_ptr(), z_warm=None if zw is None else zw.data_ptr(),
                             **{k: v.data_ptr() for k, v in out.items()})
        b.warm_curvature = int(bool(warm_curvature))
        pb = None
        if params is not None:                   # [B, n_params]: every instance its own parameters of the built-in system
            pb = torch.as_tensor(params).to(dev, torch.float64).contiguous()
            if pb.ndim != 2 or pb.shape[0] != B or pb.shape[1] != self.n_params:
                raise ValueError("params: [batch, %d] -- one row of this system's model parameters per instance, got %s" % (self.n_params, tuple(pb.shape)))
            b.params = pb.data_ptr()
        out["_keep"] = (x0, u0, zw, pb)
        return b, out

    def optimizeBatch(self, x0, u0, z_warm=None, sequences=False, stream=None, warm_curvature=False, multipliers=False, params=None):
        """`params` [B, n_params] (optional): per-instance parameters of the built-in system -- e.g. each UGV its own obstacles
        (mpcx_nlmpc_batch.params); the row layout is that of the constructor's `params`."""
        import torch
        b, out = self.make_batch(x0, u0, z_warm, sequences, warm_curvature, multipliers, params)
        s = torch.cuda.current_stream(torch.device("cuda", self.device)).cuda_stream if stream is None else stream
        check(self._lib.mpcx_nlmpc_solve_batch(self._h, C.byref(b), s))
        return out

    def time_launches(self, b, repeats, stream=None):
        import torch
        s = torch.cuda.current_stream(torch.device("cuda", self.device)).cuda_stream if stream is None else stream
        ms = C.c_float()
        check(self._lib.mpcx_nlmpc_time_solve_batch(self._h, C.byref(b), s, int(repeats), C.byref(ms)))
        return ms.value

    def optimize(self, x0, u0):
        """mpc::NLMPC::optimize(x0, lastU) (IMPC.hpp:154) through the batched kernel with B = 1; carries the previous
        solution as the next initial guess when enable_warm_start is set, as NLOptimizer does."""
        import torch
        x0 = torch.as_tensor(np.asarray(x0, float)).reshape(1, -1); u0 = torch.as_tensor(np.asarray(u0, float)).reshape(1, -1)
        zw = getattr(self, "_zprev", None) if getattr(self, "_warm", False) else None
        r = self.optimizeBatch(x0, u0, z_warm=zw, sequences=True)
        torch.cuda.synchronize()
        if int(r["status"][0]) != 3:
            self._zprev = r["z"]
        return r

    _WS_FIELDS = ("z", "d", "g", "c", "jeq", "gin", "jin", "r", "phi", "einv", "gr", "art", "br", "hinv", "mu", "glold", "s", "p",
                  "qn", "qv", "qs", "qs2", "scal", "lamw", "hook", "sp", "total")

    def debug_workspace_bytes(self):
        """size of one instance's SQP workspace in HBM"""
        fn = self._lib.mpcx_nlmpc_debug_get_ws
        fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        return 8 * check(fn(self._h, 0, None, 0, None, 0))

    def debug_workspace(self, instance):
        """testing aid: the SQP workspace of one instance after the last solve, as a dict of numpy arrays"""
        n = len(self._WS_FIELDS)
        lay = (C.c_int * n)()
        fn = self._lib.mpcx_nlmpc_debug_get_ws
        fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        total = check(fn(self._h, int(instance), None, 0, lay, n))
        buf = np.empty(total)
        check(fn(self._h, int(instance), buf.ctypes.data, total, lay, n))
        offs = list(lay)
        out = {}
        for i, name in enumerate(self._WS_FIELDS[:-1]):
            out[name] = buf[offs[i]:offs[i + 1] if i + 1 < n - 1 else total].copy()
        return out
