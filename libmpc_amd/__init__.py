"""libmpc_amd -- MI355X-native batched MPC solve engine (one hot path of libmpc++).

What is here: the linear-MPC solve path behind `mpc::LMPC<>::optimize`, for a batch of
independent instances, as a HIP kernel for gfx950 behind the C ABI of include/mpcx.h.
"""
from .lmpc import (LMPC, HorizonSlice, LParameters, Result, OptSequence, BatchResult, ResultStatus, SolutionStats, inf)
from ._capi import MpcxError

__all__ = ["LMPC", "HorizonSlice", "LParameters", "Result", "OptSequence", "BatchResult", "ResultStatus",
           "MpcxError", "SolutionStats", "inf"]
__version__ = "0.1.0"
