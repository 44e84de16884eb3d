"""Where the SQP kernel spends its cycles: per-phase shader-clock counts averaged over a batch (testing aid).
Needs the statistics build (MPCX_LIBRARY=.../libmpcx_stats.so): the product build of nlmpc_sqp_wg keeps no phase clock (round 6)."""
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tools.nlmpc_bench import make  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "ugv"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
c, x0, u0 = make(name, B)
r = c.optimizeBatch(x0, u0); torch.cuda.synchronize()
it = r["iterations"].cpu().numpy()
import os
wg = os.environ.get("MPCX_NLMPC_FORM") != "wave"
acc = np.zeros(10 if wg else 6)
qst = np.zeros(8)
dual = 0.0
for i in range(0, B, max(1, B // 16)):
    sc = c.debug_workspace(i)["scal"]
    if wg:              # the workgroup form (nlmpc_sqp_wg): ten phases, dual steps of the whole solve in slot 1
        acc += sc[2:12]; dual += sc[1] / max(1, it[i])
    else:
        acc += sc[2:8]
        qst += sc[8:16] / max(1, it[i])
qst /= len(range(0, B, max(1, B // 16)))
names = (["evaluate: cost+gradient", "evaluate: dynamics", "evaluate: constraints", "condense", "bfgs", "sub-problem", "step", "merit", "line search", "update+start"]
         if wg else ["condense", "reduce(gr,Ar,br)", "bfgs", "qp", "step+linesearch", "evaluate"])
tot = acc.sum()
for n, v in zip(names, acc):
    print("%-18s %5.1f %%" % (n, 100 * v / tot))
if wg and "stats" in os.environ.get("MPCX_LIBRARY", ""):
    raw = np.zeros(16)
    for i in range(0, B, max(1, B // 16)):
        ws = c.debug_workspace(i)
        tail = np.concatenate([ws["scal"], ws["lamw"]])[16:32]
        raw += tail / max(1, it[i])
    raw /= len(range(0, B, max(1, B // 16)))
    qn = ["x0 = -B^-1 gr", "warm: Schur complement", "warm: factor", "warm: shedding rounds + minimiser", "scan", "entering row (n, B^-1 n)", "N_W v", "solve", "N_W' r", "B^-1 w", "rest of the step"]
    print("sub-problem, cycles per SQP iteration: " + "; ".join("%s %.0f" % (n, v) for n, v in zip(qn, raw)))
    if raw[11:15].any():     # the dual part of a step, its four stretches (they are counted inside "solve" above too)
        print("   dual part: gather + zero %.0f; S^-1 t (the inverse's product / the factor's two substitutions) %.0f; z'n and ratio test %.0f; multipliers, scatter, the row's place %.0f" % tuple(raw[11:15]))
if wg and "stats" in os.environ.get("MPCX_LIBRARY", ""):
    # the Gauss-Newton start of the curvature estimate (WgSqp::init_curvature), once per solve: cycles of its parts
    ic = np.zeros(8)
    for i in range(0, B, max(1, B // 16)):
        ws = c.debug_workspace(i)
        ic += np.concatenate([ws["scal"], ws["lamw"]])[32:40]
    ic /= len(range(0, B, max(1, B // 16)))
    print("curvature start, cycles per solve: zero + the inputs' part %.0f; over the horizon: sensitivities one step on %.0f, the stage's second differences %.0f, Qx Phi %.0f, Phi' T on MFMA %.0f; "
          "tiles to the packed matrix %.0f; inversion %.0f" % tuple(ic[:7]))
if wg:
    print("dual steps per SQP iteration: %.2f" % (dual / len(range(0, B, max(1, B // 16)))))
print("iterations mean", it.mean(), " cycles per iteration (mean over sampled instances): %.0f" % (tot / len(range(0, B, max(1, B // 16))) / it.mean()))
if qst.any():       # only a build with -DMPCX_NL_STATS (make -C libmpc_amd/csrc stats; MPCX_LIBRARY=libmpc_amd/libmpcx_stats.so) fills these
    print("sub-problem, per SQP iteration: steps %.1f, inner passes %.1f, rows at the end %.1f (max %d over the solve), rows kept %.1f, shed at the warm start %.1f,"
          " warm-start cycles %.0f, factorisation cycles %.0f" % (qst[0], qst[1], qst[2], int(qst[3] * it.mean()), qst[4], qst[5], qst[6], qst[7]))
