"""Throughput of the run-time compiled hook route against the built-in model (Van der Pol example, batch 4096): testing aid."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
import torch
from libmpc_amd.nlmpc import NLMPC, NLParameters, VANDERPOL

SRC = dict(state_fn="dx(0) = ((1.0 - (x(1) * x(1))) * x(0)) - x(1) + u(0); dx(1) = x(0);",
           objective_fn="return x.array().square().sum() + u.array().square().sum();",
           ineq_fn="for (int i = 0; i < ineq_c; i++) { in_con(i) = u(i, 0) - 0.5; }")
B = 4096
rng = np.random.default_rng(0)
x0 = torch.from_numpy(rng.uniform(-1, 1, size=(B, 2))); u0 = torch.zeros(B, 1, dtype=torch.float64)
t0 = time.perf_counter()
usr = NLMPC.from_sources(2, 1, 2, 10, 5, 11, 0, 0.1, **SRC)
t_jit = time.perf_counter() - t0
zoo = NLMPC(VANDERPOL, 10, 5, 0.1)
for name, c in (("built-in model", zoo), ("hooks compiled at run time", usr)):
    c.setOptimizerParameters(NLParameters(maximum_iteration=200))
    r = c.optimizeBatch(x0, u0); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        r = c.optimizeBatch(x0, u0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    print("%-28s %8.2f ms per batch of %d = %9.0f solves/s   converged %.3f  mean iterations %.1f" %
          (name, dt * 1e3, B, B / dt, (r["status"] == 0).float().mean().item(), r["iterations"].float().mean().item()))
print("run-time compilation of the hooks: %.1f s" % t_jit)
