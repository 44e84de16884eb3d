"""pympcxx -- the name and surface of libmpc++'s Python module (reference python/pybind_export.cpp:13-213), served by the
MI355X engine in libmpc_amd.  Code written against the reference's binding (`import pympcxx`) runs against this package
for the solve path: the same class names, constructors, setter overloads, enums with their exported values, parameter
objects with the reference's field names.

What is different, and why
  * `NLMPC.setStateSpaceFunction / setObjectiveFunction / setIneqConFunction / setEqConFunction / setOutputFunction` take the
    hook as C++ TEXT -- the body of the lambda one would write against the C++ API -- not a Python callable: the hooks run
    inside a GPU kernel (compiled at run time for gfx950), where a Python function cannot.  Parameter names and scope are
    those of `libmpc_amd.NLMPC.from_sources`.  A callable raises a TypeError that says so.
  * every controller also has `optimizeBatch(...)`: B instances per launch (libmpc_amd.LMPC / NLMPC).
There is no CPU fallback: without libmpcx.so and an MI355X, solving raises.
"""
from __future__ import annotations

import numpy as np

from libmpc_amd.lmpc import (LMPC, HorizonSlice, LParameters, OptSequence, Result, ResultStatus, SolutionStats, inf)  # noqa: F401
from libmpc_amd.nlmpc import NLMPC as _BatchedNLMPC
from libmpc_amd.nlmpc import NLParameters  # noqa: F401

Parameters = LParameters            # mpc::Parameters is the common base (maximum_iteration, time_limit, enable_warm_start)


class LoggerLevel:                  # py::enum_<mpc::Logger::LogLevel> with export_values() (pybind_export.cpp:163-168)
    DEEP, NORMAL, ALERT, NONE = range(4)


DEEP, NORMAL, ALERT, NONE = LoggerLevel.DEEP, LoggerLevel.NORMAL, LoggerLevel.ALERT, LoggerLevel.NONE
# py::enum_<mpc::ResultStatus> ... export_values() (pybind_export.cpp:193-199)
UNKNOWN, SUCCESS, MAX_ITERATION, INFEASIBLE, ERROR = (ResultStatus.UNKNOWN, ResultStatus.SUCCESS, ResultStatus.MAX_ITERATION,
                                                      ResultStatus.INFEASIBLE, ResultStatus.ERROR)


class NLMPC:
    """pympcxx.NLMPC(nx, nu, ny, ph, ch, ineq, eq) (pybind_export.cpp:59-84).  The device controller is built by the first
    optimize() after the hooks or the sampling time changed."""

    def __init__(self, nx, nu, ny, ph, ch, ineq, eq, device=0):
        self._dims = tuple(int(v) for v in (nx, nu, ny, ph, ch, ineq, eq))
        self._device = device
        self._ts = 0.0
        self._src = {}
        self._prm = NLParameters()
        self._bounds = []
        self._scale = {}
        self._c = None
        self._last = Result(cmd=np.zeros(self._dims[1]))
        self._seq = None
        self._stats = SolutionStats()

    # ---- hooks -------------------------------------------------------------------------------------------------------
    def _hook(self, key, src):
        if callable(src) or not isinstance(src, str):
            raise TypeError("the hook runs inside a GPU kernel: pass the C++ body of the lambda as a string "
                            "(see libmpc_amd.NLMPC.from_sources for the parameter names), not a Python callable")
        self._src[key] = src
        self._c = None
        return True

    def setStateSpaceFunction(self, src, eq_tol=1e-10):
        return self._hook("state_fn", src)

    def setObjectiveFunction(self, src):
        return self._hook("objective_fn", src)

    def setIneqConFunction(self, src, ineq_tol=1e-10):
        return self._hook("ineq_fn", src)

    def setEqConFunction(self, src, eq_tol=1e-10):
        return self._hook("eq_fn", src)

    def setOutputFunction(self, src):
        return self._hook("output_fn", src)

    def setPreamble(self, src):
        """extension: C++ text pasted before the hooks at namespace scope (constants, helper __device__ functions)"""
        return self._hook("preamble", src)

    # ---- the rest of the reference surface ------------------------------------------------------------------------------
    def setDiscretizationSamplingTime(self, ts):
        self._ts = float(ts)
        self._c = None
        return True

    def setInputScale(self, scaling):
        self._scale["input"] = np.asarray(scaling, float)
        if self._c is not None:
            self._c.setInputScale(self._scale["input"])

    def setStateScale(self, scaling):
        self._scale["state"] = np.asarray(scaling, float)
        if self._c is not None:
            self._c.setStateScale(self._scale["state"])

    def setOptimizerParameters(self, params):
        self._prm = params
        if self._c is not None:
            self._c.setOptimizerParameters(params)

    def setLoggerLevel(self, level):
        return True

    def setLoggerPrefix(self, prefix):
        return True

    def _bound(self, which, lo, hi, slice_=None):
        self._bounds.append((which, np.asarray(lo, float), np.asarray(hi, float), slice_))
        if self._c is not None:
            return getattr(self._c, which)(lo, hi, slice_)
        s = None if slice_ is None else ((slice_.start, slice_.end) if hasattr(slice_, "start") else tuple(slice_))
        horizon = self._dims[3] if which == "setStateBounds" else self._dims[4]
        return s is None or s == (-1, -1) or (0 <= s[0] < s[1] <= horizon)

    def setStateBounds(self, lo, hi, slice_=None):
        return self._bound("setStateBounds", lo, hi, slice_)

    def setInputBounds(self, lo, hi, slice_=None):
        return self._bound("setInputBounds", lo, hi, slice_)

    def setOutputBounds(self, *_a, **_k):
        raise RuntimeError("Output constraints cannot be set for this type of MPC")        # NLMPC.hpp:318-325

    def _controller(self):
        if self._c is None:
            if "state_fn" not in self._src or "objective_fn" not in self._src:
                raise RuntimeError("set the state-space and the objective function first")
            nx, nu, ny, ph, ch, ineq, eq = self._dims
            c = _BatchedNLMPC.from_sources(nx, nu, ny, ph, ch, ineq, eq, self._ts, device=self._device, **self._src)
            c.setOptimizerParameters(self._prm)
            if "input" in self._scale:
                c.setInputScale(self._scale["input"])
            if "state" in self._scale:
                c.setStateScale(self._scale["state"])
            for which, lo, hi, sl in self._bounds:
                getattr(c, which)(lo, hi, sl)
            self._c = c
        return self._c

    def optimize(self, x0, lastU):
        import time
        t0 = time.perf_counter()
        r = self._controller().optimize(x0, lastU)
        st = int(r["status"][0])
        res = Result(solver_status=int(r["solver_status"][0]), is_feasible=bool(int(r["is_feasible"][0])),
                     solver_status_msg="", cost=float(r["cost"][0]), status=st, cmd=r["cmd"][0].cpu().numpy().copy())
        self._seq = OptSequence(state=r["seq_state"][0].cpu().numpy(), output=r["seq_output"][0].cpu().numpy(),
                                input=r["seq_input"][0].cpu().numpy())
        self._last = res
        self._stats.add(time.perf_counter() - t0, st)
        return res

    def optimizeBatch(self, x0, lastU, **kw):
        return self._controller().optimizeBatch(x0, lastU, **kw)

    def getLastResult(self):
        return self._last

    def getOptimalSequence(self):
        return self._seq

    def getExecutionStats(self):
        return self._stats

    def resetStats(self):
        self._stats = SolutionStats()


__all__ = ["LMPC", "NLMPC", "Parameters", "LParameters", "NLParameters", "LoggerLevel", "Result", "SolutionStats", "ResultStatus",
           "HorizonSlice", "OptSequence", "DEEP", "NORMAL", "ALERT", "NONE", "UNKNOWN", "SUCCESS", "MAX_ITERATION", "INFEASIBLE",
           "ERROR", "inf"]
