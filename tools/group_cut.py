"""Timing experiment: lmpc_solve_group cut short (libraries built by tools/group_cut.sh with -DMPCX_GROUP_CUT=k: the kernel returns at its entry / after its
inputs are staged / after the first product / after the second) -- what each part of the assemble phase costs a launch, measured without stamps in the way."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc
ph, B = 20, 4096
c = quadrotor_lmpc(ph, device=0)
c.debug_use_fused(2)
x0, u0, yref = quadrotor_batch(B)
batch, res, keep = c.make_batch(x0, u0, yref=yref)
for _ in range(3):
    c.launch(batch)
torch.cuda.synchronize()
print("step ms %.4f (lmpc_solve_group + the idle fallback launch, HIP events over 200 steps)" % c.time_launches(batch, 200))
