"""CPU baseline of the non-linear path (a script, not a test; lives under tests/ because it runs the oracle): the oracle
-- numpy restatement of the reference's callbacks + scipy SLSQP -- on the first instances of the synthetic batches that
tools/nlmpc_bench.py times on the GPU.  Usage: python tests/nlmpc_cpu_baseline.py [ugv|vanderpol|osc6|osc8] [instances]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nlmpc_numpy as ref  # noqa: E402


def batch(name, B, seed=0):
    """same inputs as tools/nlmpc_bench.py::make"""
    rng = np.random.default_rng(seed)
    if name == "ugv":
        x0 = np.zeros((B, 4)); x0[:, :2] = rng.uniform(-0.5, 0.5, size=(B, 2)); nu = 2
    elif name == "vanderpol":
        x0 = rng.uniform(-1, 1, size=(B, 2)); nu = 1
    else:
        n = int(name[3:])
        x0 = rng.uniform(-0.1, 0.1, size=(B, 2 * n)); x0[:, 0] += 1.0; nu = n
    return x0, np.zeros((B, nu))


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "ugv"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    m = dict(ugv=lambda: ref.ugv(30, 30), vanderpol=lambda: ref.vanderpol(10, 5, 0.1), osc6=lambda: ref.oscillators(6, 20, 10),
             osc8=lambda: ref.oscillators(8, 30, 15))[name]()
    x0, u0 = batch(name, 4096 if name in ("ugv", "vanderpol") else 1024)
    t0 = time.perf_counter()
    for i in range(n):
        m.solve(x0[i], u0[i], max_iter=150, hard=(name != "ugv"))
    dt = time.perf_counter() - t0
    print(json.dumps(dict(workload=name, value=n / dt, unit="solves/s", cores=1, kind="port",
                          sample="first %d instances, numpy callbacks + scipy SLSQP" % n)))
