"""libmpc_amd -- MI355X-native batched MPC solve engine (the hot paths of libmpc++).

What is here: the solve paths behind `mpc::LMPC<>::optimize` and `mpc::NLMPC<>::optimize`, for a batch of independent
instances of one controller, as HIP kernels for gfx950 behind the C ABI of include/mpcx.h.  `LMPC` / `NLMPC` mirror the
reference's front-ends (ctypes; PyTorch only provides device memory and streams), `utils.discretization` its c2d helper.
There is no CPU fallback: a missing libmpcx.so or GPU raises.
"""
from .lmpc import (LMPC, HorizonSlice, LParameters, Result, OptSequence, BatchResult, ResultStatus, SolutionStats, inf)
from .nlmpc import NLMPC, NLMPCEvaluator, NLParameters
from ._capi import MpcxError
from .bank import LMPCBank, LMPCHetero, group_by_model

__all__ = ["LMPC", "NLMPC", "NLMPCEvaluator", "HorizonSlice", "LParameters", "NLParameters", "Result", "OptSequence",
           "BatchResult", "ResultStatus", "MpcxError", "SolutionStats", "inf", "LMPCBank", "LMPCHetero", "group_by_model"]
__version__ = "0.1.0"
