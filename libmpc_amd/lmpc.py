"""Host-side mirror of libmpc++'s linear MPC front-end for the batched MI355X engine.

`LMPC` keeps the method names, argument meaning and return/raise behaviour of
`mpc::LMPC<>` (reference include/mpc/LMPC.hpp, and its pybind export
python/pybind_export.cpp:59-123) for the one path this package replaces -- what sits
behind `optimize()` -- and adds `optimizeBatch()`: B independent instances of the same
controller solved by one HIP kernel launch through the C ABI in include/mpcx.h.

PyTorch is plumbing here (device buffers and streams), not the product: all numerics
run in libmpcx.so.  There is no CPU fallback; a missing library or GPU raises.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _capi
from ._capi import LParams, MpcxError, check

inf = float("inf")


class HorizonSlice:
    """Half-open step range [start, end); {-1,-1} = whole horizon (Types.hpp:57-82)."""

    def __init__(self, start, end):
        self.start, self.end = int(start), int(end)

    @staticmethod
    def all():
        return HorizonSlice(-1, -1)


def _slice(s):
    if s is None:
        return HorizonSlice.all()
    if isinstance(s, HorizonSlice):
        return s
    a, b = s
    return HorizonSlice(a, b)


class SolutionStats:
    """mpc::SolutionStats (Profiler.hpp:19-60) as the pybind module exposes it; times in seconds"""

    def __init__(self):
        self.totalSolutionTime = 0.0
        self.numberOfSolutions = 0
        self.minSolutionTime = float("inf")
        self.maxSolutionTime = 0.0
        self.averageSolutionTime = 0.0
        self.standardDeviation = 0.0
        self.solutionsStates = {}
        self._sq = 0.0

    def add(self, dt, status):
        self.totalSolutionTime += dt
        self.numberOfSolutions += 1
        self.minSolutionTime = min(self.minSolutionTime, dt)
        self.maxSolutionTime = max(self.maxSolutionTime, dt)
        self._sq += dt * dt
        n = self.numberOfSolutions
        self.averageSolutionTime = self.totalSolutionTime / n
        self.standardDeviation = max(self._sq / n - self.averageSolutionTime ** 2, 0.0) ** 0.5
        self.solutionsStates[status] = self.solutionsStates.get(status, 0) + 1


class ResultStatus:
    SUCCESS, MAX_ITERATION, INFEASIBLE, ERROR, UNKNOWN = range(5)


def LParameters(**kw) -> LParams:
    """mpc::LParameters with the reference defaults (Types.hpp:146-161)."""
    p = LParams()
    _capi.lib().mpcx_lparams_default(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


@dataclass
class Result:
    """mpc::Result<nu> (Types.hpp:168-182)."""
    solver_status: int = 0
    is_feasible: bool = False
    solver_status_msg: str = ""
    cost: float = 0.0
    status: int = ResultStatus.UNKNOWN
    cmd: np.ndarray = field(default_factory=lambda: np.zeros(0))


@dataclass
class OptSequence:
    """mpc::OptSequence (Types.hpp:184-198): row i = horizon step i."""
    state: np.ndarray
    output: np.ndarray
    input: np.ndarray


@dataclass
class BatchResult:
    """Result<nu> for B instances as structure-of-arrays of device tensors."""
    cmd: "object"
    cost: "object"
    status: "object"
    solver_status: "object"
    is_feasible: "object"
    iterations: "object"
    active_lower: "object" = None
    active_upper: "object" = None
    seq_state: "object" = None
    seq_output: "object" = None
    seq_input: "object" = None
    polish_rounds: "object" = None
    active_count: "object" = None


def _cm(a, rows, cols):
    """column-major float64 host copy with a shape check"""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1 and cols == 1:
        a = a.reshape(rows, 1)
    if a.shape != (rows, cols):
        raise ValueError(f"expected shape {(rows, cols)}, got {a.shape}")
    return np.asfortranarray(a)


def _vec(a, n):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
    if a.shape[0] != n:
        raise ValueError(f"expected length {n}, got {a.shape[0]}")
    return a


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class LMPC:
    """Linear MPC controller whose optimize() runs on an MI355X.

    Mirrors mpc::LMPC<Tnx,Tnu,Tndu,Tny,Tph,Tch> with run-time sizes
    (the reference's MPC_DYNAMIC form, LMPC.hpp:57-62).  `device=-1` builds a
    host-only handle: setters and the condensing work, any solve raises.
    """

    def __init__(self, nx, nu, ndu, ny, ph, ch, device=0):
        self._lib = _capi.lib()
        self.nx, self.nu, self.ndu, self.ny, self.ph, self.ch = map(int, (nx, nu, ndu, ny, ph, ch))
        self.device = int(device)
        d = _capi.Dims(self.nx, self.nu, self.ndu, self.ny, self.ph, self.ch)
        self._h = C.c_void_p()
        check(self._lib.mpcx_lmpc_create(C.byref(d), self.device, C.byref(self._h)))
        self._last = Result(cmd=np.zeros(self.nu))
        self._stats = SolutionStats()
        self._last_u0 = np.zeros(self.nu)
        self._seq = OptSequence(np.zeros((self.ph + 1, self.nx)), np.zeros((self.ph + 1, self.ny)),
                                np.zeros((self.ph + 1, self.nu)))

    def __del__(self):
        try:
            if self._h:
                self._lib.mpcx_lmpc_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- not available on the linear front-end (LMPC.hpp:68-100) -------------------
    def setDiscretizationSamplingTime(self, ts):
        raise RuntimeError("Linear MPC supports only discrete time systems")

    def setInputScale(self, scaling):
        raise RuntimeError("Linear MPC does not support input scaling")

    def setStateScale(self, scaling):
        raise RuntimeError("Linear MPC does not support state scaling")

    def setLoggerLevel(self, level):
        return True

    def setLoggerPrefix(self, prefix):
        return True

    # -- set-up ---------------------------------------------------------------------
    def _ok(self, rc):
        if rc == _capi.E_INVALID:
            return False          # the reference's setters return false on a bad slice
        check(rc)
        return True

    def setOptimizerParameters(self, params: LParams):
        check(self._lib.mpcx_lmpc_set_optimizer_parameters(self._h, C.byref(params)))
        self._warm_enabled = bool(params.enable_warm_start)
        self._warm_prev = None

    def setStrictInfeasibility(self, on=True):
        """extension: report infeasible QPs as INFEASIBLE / NaN instead of the reference's
        MAX_ITERATION + last iterate (see include/mpcx.h)"""
        check(self._lib.mpcx_lmpc_set_strict_infeasibility(self._h, int(bool(on))))

    def setStateSpaceModel(self, A, B, Cm):
        A, B, Cm = _cm(A, self.nx, self.nx), _cm(B, self.nx, self.nu), _cm(Cm, self.ny, self.nx)
        return self._ok(self._lib.mpcx_lmpc_set_state_space_model(self._h, _p(A), _p(B), _p(Cm)))

    def setDisturbances(self, Bd, Dd):
        Bd, Dd = _cm(Bd, self.nx, self.ndu), _cm(Dd, self.ny, self.ndu)
        return self._ok(self._lib.mpcx_lmpc_set_disturbances(self._h, _p(Bd), _p(Dd)))

    def setObjectiveWeights(self, OWeight, UWeight, DeltaUWeight, slice=None):
        ow = np.asarray(OWeight, dtype=np.float64)
        if ow.ndim == 2 and slice is None:
            O, U, D = _cm(OWeight, self.ny, self.ph), _cm(UWeight, self.nu, self.ph), _cm(DeltaUWeight, self.nu, self.ph)
            return self._ok(self._lib.mpcx_lmpc_set_objective_weights(self._h, _p(O), _p(U), _p(D)))
        s = _slice(slice)
        o, u, d = _vec(OWeight, self.ny), _vec(UWeight, self.nu), _vec(DeltaUWeight, self.nu)
        return self._ok(self._lib.mpcx_lmpc_set_objective_weights_slice(self._h, _p(o), _p(u), _p(d), s.start, s.end))

    def _bounds(self, lo, hi, rows, cols, fmat, fslice, slice):
        a = np.asarray(lo, dtype=np.float64)
        if a.ndim == 2 and slice is None:
            L, H = _cm(lo, rows, cols), _cm(hi, rows, cols)
            return self._ok(fmat(self._h, _p(L), _p(H)))
        s = _slice(slice)
        l, h = _vec(lo, rows), _vec(hi, rows)
        return self._ok(fslice(self._h, _p(l), _p(h), s.start, s.end))

    def setStateBounds(self, XMin, XMax, slice=None):
        return self._bounds(XMin, XMax, self.nx, self.ph, self._lib.mpcx_lmpc_set_state_bounds,
                            self._lib.mpcx_lmpc_set_state_bounds_slice, slice)

    def setInputBounds(self, UMin, UMax, slice=None):
        return self._bounds(UMin, UMax, self.nu, self.ch, self._lib.mpcx_lmpc_set_input_bounds,
                            self._lib.mpcx_lmpc_set_input_bounds_slice, slice)

    def setOutputBounds(self, YMin, YMax, slice=None):
        return self._bounds(YMin, YMax, self.ny, self.ph, self._lib.mpcx_lmpc_set_output_bounds,
                            self._lib.mpcx_lmpc_set_output_bounds_slice, slice)

    def setScalarConstraint(self, *args):
        """(min, max, X, U, slice) as LMPC.hpp:355 or (index, min, max, X, U) as LMPC.hpp:409."""
        if len(args) != 5:
            raise TypeError("setScalarConstraint takes (min, max, X, U, slice) or (index, min, max, X, U)")
        if isinstance(args[4], (HorizonSlice, tuple, list)) or args[4] is None:
            smin, smax, X, U, sl = args
            s = _slice(sl)
            X, U = _vec(X, self.nx), _vec(U, self.nu)
            return self._ok(self._lib.mpcx_lmpc_set_scalar_constraint_slice(
                self._h, float(smin), float(smax), _p(X), _p(U), s.start, s.end))
        index, smin, smax, X, U = args
        X, U = _vec(X, self.nx), _vec(U, self.nu)
        return self._ok(self._lib.mpcx_lmpc_set_scalar_constraint_index(
            self._h, int(index), float(smin), float(smax), _p(X), _p(U)))

    def setReferences(self, outRef, cmdRef, deltaCmdRef, slice=None):
        a = np.asarray(outRef, dtype=np.float64)
        if a.ndim == 2 and slice is None:
            Y, U, D = _cm(outRef, self.ny, self.ph), _cm(cmdRef, self.nu, self.ph), _cm(deltaCmdRef, self.nu, self.ph)
            return self._ok(self._lib.mpcx_lmpc_set_references(self._h, _p(Y), _p(U), _p(D)))
        s = _slice(slice)
        y, u, d = _vec(outRef, self.ny), _vec(cmdRef, self.nu), _vec(deltaCmdRef, self.nu)
        return self._ok(self._lib.mpcx_lmpc_set_references_slice(self._h, _p(y), _p(u), _p(d), s.start, s.end))

    def setExogenousInputs(self, uMeas, slice=None):
        a = np.asarray(uMeas, dtype=np.float64)
        if a.ndim == 2 and slice is None:
            Dm = _cm(uMeas, self.ndu, self.ph)
            return self._ok(self._lib.mpcx_lmpc_set_exogenous_inputs(self._h, _p(Dm)))
        s = _slice(slice)
        d = _vec(uMeas, self.ndu)
        return self._ok(self._lib.mpcx_lmpc_set_exogenous_inputs_slice(self._h, _p(d), s.start, s.end))

    # -- introspection -----------------------------------------------------------------
    def setup(self):
        check(self._lib.mpcx_lmpc_setup(self._h))

    def info(self):
        i = _capi.Info()
        check(self._lib.mpcx_lmpc_get_info(self._h, C.byref(i)))
        return {n: getattr(i, n) for n, _ in _capi.Info._fields_}

    def debug_get(self, name):
        """testing aid: condensed arrays as computed by the host set-up"""
        f = self._lib.mpcx_lmpc_debug_get
        n = check(f(self._h, name.encode(), None, 0))
        out = np.zeros(n)
        check(f(self._h, name.encode(), _p(out), n))
        return out

    def setTotalBatch(self, total):
        """sharding: this controller is given contiguous shards of a batch of `total` instances (one rank of N); the kernel form is then chosen
        for the whole batch's size and a shard's results are bit for bit the rows of the unsharded solve (0: every call is a whole batch)"""
        check(self._lib.mpcx_lmpc_set_total_batch(self._h, int(total)))
        return True

    def debug_force_generic(self, on=True):
        """testing aid: route every batch through the generic (roll-out) assemble kernel"""
        check(self._lib.mpcx_lmpc_debug_force_generic(self._h, int(bool(on))))

    def debug_use_fused(self, on=True):
        """experiment / testing knob: False / 0 = assemble and solve as two kernels, True / 1 = the record computed inside the solve
        kernel by one mat-vec (persistent form from 1024 instances on), 2 = assemble + solve in one workgroup of sixteen
        wavefronts (lmpc_solve_group), -1 = automatic (the default: the workgroup form up to 4096 instances, two kernels beyond; the fused
        forms only on request)"""
        check(self._lib.mpcx_lmpc_debug_use_fused(self._h, -1 if on is None else int(on)))

    def _torch(self):
        import torch
        if self.device < 0 or not torch.cuda.is_available():
            raise MpcxError(_capi.E_DEVICE, "optimize needs an MI355X: libmpc_amd has no CPU solve path")
        return torch, torch.device("cuda", self.device)

    def _dev(self, torch, dev, a, shape):
        if a is None:
            return None
        t = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a, dtype=np.float64))
        t = t.to(device=dev, dtype=torch.float64).contiguous()
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return t

    def _ref(self, torch, dev, a, B, n):
        """classify a per-solve reference: None -> shared, [B,n] -> per instance, [B,ph,n] -> per step"""
        if a is None:
            return None, _capi.REF_SHARED
        t = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a, dtype=np.float64))
        t = t.to(device=dev, dtype=torch.float64).contiguous()
        if tuple(t.shape) == (B, n):
            return t, _capi.REF_PER_INSTANCE
        if tuple(t.shape) == (B, self.ph, n):
            return t, _capi.REF_PER_STEP
        raise ValueError(f"reference must be [B,{n}] or [B,{self.ph},{n}], got {tuple(t.shape)}")

    def make_batch(self, x0, u0, yref=None, uref=None, duref=None, dmeas=None,
                   want_active=False, want_sequence=False, warm=None, warm_shift=False):
        """Allocate outputs and fill the mpcx_lmpc_batch descriptor.  Returns (Batch, BatchResult, keepalive)."""
        torch, dev = self._torch()
        x0t = x0 if hasattr(x0, "shape") else np.asarray(x0)
        B = int(x0t.shape[0])
        x0 = self._dev(torch, dev, x0, (B, self.nx))
        u0 = self._dev(torch, dev, u0, (B, self.nu))
        yr, ym = self._ref(torch, dev, yref, B, self.ny)
        ur, um = self._ref(torch, dev, uref, B, self.nu)
        dr, dmo = self._ref(torch, dev, duref, B, self.nu)
        de, dem = self._ref(torch, dev, dmeas, B, self.ndu)
        i = self.info()
        f64, i32 = torch.float64, torch.int32
        res = BatchResult(
            cmd=torch.empty((B, self.nu), dtype=f64, device=dev),
            cost=torch.empty((B,), dtype=f64, device=dev),
            status=torch.empty((B,), dtype=i32, device=dev),
            solver_status=torch.empty((B,), dtype=i32, device=dev),
            is_feasible=torch.empty((B,), dtype=i32, device=dev),
            iterations=torch.empty((B,), dtype=i32, device=dev),
            polish_rounds=torch.zeros((B,), dtype=i32, device=dev),
            active_count=torch.zeros((B,), dtype=i32, device=dev))
        if want_active:
            res.active_lower = torch.zeros((B, i["active_words"]), dtype=i32, device=dev)
            res.active_upper = torch.zeros((B, i["active_words"]), dtype=i32, device=dev)
        if want_sequence:
            res.seq_state = torch.empty((B, self.ph + 1, self.nx), dtype=f64, device=dev)
            res.seq_output = torch.empty((B, self.ph + 1, self.ny), dtype=f64, device=dev)
            res.seq_input = torch.empty((B, self.ph + 1, self.nu), dtype=f64, device=dev)

        def ptr(t):
            return None if t is None else C.c_void_p(t.data_ptr())

        b = _capi.Batch()
        b.batch = B
        b.x0, b.u0 = ptr(x0), ptr(u0)
        b.yref, b.yref_mode = ptr(yr), ym
        b.uref, b.uref_mode = ptr(ur), um
        b.duref, b.duref_mode = ptr(dr), dmo
        b.dmeas, b.dmeas_mode = ptr(de), dem
        b.cmd, b.cost = ptr(res.cmd), ptr(res.cost)
        b.status, b.solver_status = ptr(res.status), ptr(res.solver_status)
        b.is_feasible, b.iterations = ptr(res.is_feasible), ptr(res.iterations)
        b.active_lower, b.active_upper = ptr(res.active_lower), ptr(res.active_upper)
        b.seq_state, b.seq_output, b.seq_input = ptr(res.seq_state), ptr(res.seq_output), ptr(res.seq_input)
        b.polish_rounds, b.active_count = ptr(res.polish_rounds), ptr(res.active_count)
        wl = wu = None
        if warm is not None:                       # a previous BatchResult (with its active sets) or a (lower, upper) pair
            wl, wu = (warm.active_lower, warm.active_upper) if hasattr(warm, "active_lower") else warm
            if wl is None or wu is None:
                raise ValueError("warm start needs the previous solve's active sets (want_active=True)")
            wl = wl.to(dev).contiguous(); wu = wu.to(dev).contiguous()
            if tuple(wl.shape) != (B, i["active_words"]) or tuple(wu.shape) != (B, i["active_words"]):
                raise ValueError("warm-start active sets have the wrong shape")
            b.warm_active_lower, b.warm_active_upper = ptr(wl), ptr(wu)
            b.warm_shift = int(bool(warm_shift))
        keep = (x0, u0, yr, ur, dr, de, wl, wu)
        return b, res, keep

    def launch(self, batch, stream=None, keep=None):
        """One asynchronous kernel launch for a descriptor built by make_batch().

        A handle owns one workspace and one set of dispatch queues: keep ONE launch of a controller in flight at a time
        (launches on the same stream are ordered; for overlapping batches use one controller per stream, as bench.py
        does).  make_batch() fills its tensors on torch's current stream: a different launch stream first waits for it,
        and `keep` (the tensors the descriptor points at) is tied to the launch stream so that the caching allocator
        does not recycle them while the kernels still read them."""
        torch, _ = self._torch()
        cur = torch.cuda.current_stream(self.device)
        s = stream if stream is not None else cur
        if s.cuda_stream != cur.cuda_stream:
            s.wait_stream(cur)
            for t in (keep or ()):
                if t is not None:
                    t.record_stream(s)
        check(self._lib.mpcx_lmpc_solve_batch(self._h, C.byref(batch), C.c_void_p(s.cuda_stream)))

    def make_graph(self, batch, stream):
        """One step as a HIP graph (mpcx_lmpc_graph_create): the launches of `launch(batch)` captured once on `stream` (a
        non-default torch stream) and replayed by `launch_graph`.  The descriptor's tensors stay where they are; write new
        inputs into them in place."""
        g = C.c_void_p()
        check(self._lib.mpcx_lmpc_graph_create(self._h, C.byref(batch), C.c_void_p(stream.cuda_stream), C.byref(g)))
        return g

    def launch_graph(self, graph, stream):
        check(self._lib.mpcx_lmpc_graph_launch(graph, C.c_void_p(stream.cuda_stream)))

    def destroy_graph(self, graph):
        check(self._lib.mpcx_lmpc_graph_destroy(graph))

    def time_launches(self, batch, repeats, stream=None):
        """Mean kernel time (ms) over `repeats` launches, HIP events on the launch stream."""
        torch, _ = self._torch()
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        ms = C.c_float()
        check(self._lib.mpcx_lmpc_time_solve_batch(self._h, C.byref(batch), C.c_void_p(s.cuda_stream), int(repeats), C.byref(ms)))
        return ms.value

    def optimizeBatch(self, x0, lastU, yref=None, uref=None, duref=None, dmeas=None,
                      want_active=False, want_sequence=False, stream=None, warm=None, warm_shift=False) -> BatchResult:
        """B independent LOptimizer::run calls (LOptimizer.hpp:189) in one launch.  `warm`: the BatchResult of the
        previous control tick (solved with want_active=True) -- its active sets seed the working sets; warm_shift=True
        looks each row up one horizon step later (receding horizon)."""
        b, res, keep = self.make_batch(x0, lastU, yref, uref, duref, dmeas, want_active or warm is not None, want_sequence, warm, warm_shift)
        self.launch(b, stream, keep)
        res._inputs = keep                     # the inputs live as long as the result that was computed from them
        return res

    def optimize(self, x0, lastU) -> Result:
        """IMPC::optimize (IMPC.hpp:149-166) for one instance, through the batched path."""
        import time
        t_start = time.perf_counter()
        self._last_u0 = np.asarray(lastU, dtype=np.float64).reshape(self.nu).copy()
        warm = getattr(self, "_warm_prev", None) if getattr(self, "_warm_enabled", False) else None
        r = self.optimizeBatch(np.asarray(x0, dtype=np.float64).reshape(1, self.nx),
                               np.asarray(lastU, dtype=np.float64).reshape(1, self.nu), want_sequence=True,
                               want_active=getattr(self, "_warm_enabled", False), warm=warm)
        if getattr(self, "_warm_enabled", False):
            self._warm_prev = r                   # LOptimizer keeps optimal_prev_x / _y (LOptimizer.hpp:295-296, 372)
        sst = int(r.solver_status[0].item())
        self._last = Result(solver_status=sst, is_feasible=bool(r.is_feasible[0].item()), solver_status_msg="",
                            cost=float(r.cost[0].item()), status=int(r.status[0].item()),
                            cmd=r.cmd[0].cpu().numpy().copy())
        self._seq = OptSequence(r.seq_state[0].cpu().numpy().copy(), r.seq_output[0].cpu().numpy().copy(),
                                r.seq_input[0].cpu().numpy().copy())
        self._stats.add(time.perf_counter() - t_start, self._last.status)
        return self._last

    def getLastResult(self) -> Result:
        return self._last

    def getOptimalSequence(self) -> OptSequence:
        return self._seq

    # ---- the rest of the pybind module's LMPC surface (python/pybind_export.cpp:93-123) --------------------------
    def getExecutionStats(self) -> "SolutionStats":
        """IMPC::getExecutionStats (IMPC.hpp:201-204, Profiler.hpp): wall time of the optimize() calls made so far."""
        return self._stats

    def resetStats(self):
        self._stats = SolutionStats()

    def getSolverWarmStartDual(self):
        """LMPC::getSolverWarmStartDual (LMPC.hpp:697-700) returns OSQP's dual vector; what this engine carries between
        ticks is the active set, so the vector holds -1 / 0 / +1 per reference row (lower / inactive / upper): the
        sign pattern of OSQP's y."""
        r = getattr(self, "_warm_prev", None)
        i = self.info()
        y = np.zeros(i["m_ref"])
        if r is not None:
            al, au = (r.active_lower, r.active_upper) if hasattr(r, "active_lower") else r
            lo = al[0].cpu().numpy().view(np.uint32); hi = au[0].cpu().numpy().view(np.uint32)
            for k in range(i["m_ref"]):
                if (lo[k >> 5] >> (k & 31)) & 1:
                    y[k] = -1.0
                elif (hi[k >> 5] >> (k & 31)) & 1:
                    y[k] = 1.0
        return y

    def getSolverWarmStartPrimal(self):
        """LMPC::getSolverWarmStartPrimal (LMPC.hpp:677-680): the reference QP's primal vector [xi_0..xi_ph | du_0..du_ph-1]
        (ProblemBuilder.hpp:70-76), rebuilt from the last optimal sequence."""
        seq = self._seq
        ph, nx, nu = self.ph, self.nx, self.nu
        v = np.vstack([self._last_u0[None, :], seq.input[:ph]])              # v_i = u_{i-1}
        xi = np.hstack([seq.state, v])
        du = np.diff(v, axis=0)
        return np.concatenate([xi.reshape(-1), du.reshape(-1)])

    def setSolverWarmStart(self, warm_primal, warm_dual):
        """LMPC::setSolverWarmStart (LMPC.hpp:716-722): only the sign pattern of the dual is used (see above)."""
        import torch
        i = self.info()
        y = np.asarray(warm_dual, dtype=np.float64).reshape(-1)
        if y.size != i["m_ref"]:
            raise ValueError("dual vector must have one entry per reference constraint row")
        lo = np.zeros(i["active_words"], dtype=np.uint32); hi = np.zeros(i["active_words"], dtype=np.uint32)
        for k in np.nonzero(y)[0]:
            (lo if y[k] < 0 else hi)[k >> 5] |= np.uint32(1 << (k & 31))
        dev = torch.device("cuda", self.device)
        self._warm_prev = (torch.from_numpy(lo.view(np.int32)[None, :]).to(dev), torch.from_numpy(hi.view(np.int32)[None, :]).to(dev))
