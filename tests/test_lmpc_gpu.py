"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

Tolerance: BASELINE.json asks for u* within 1e-5 relative of the reference CPU path and
bit-exact active-set indices.  The oracle's polished solution is the reference point; where
its polish did not succeed (ADMM-accuracy iterate, eps 1e-4) the comparison is loosened and
counted."""
import numpy as np
import pytest

from helpers import bits_to_rows, quadrotor_oracle

pytestmark = pytest.mark.gpu

RTOL_CMD = 1e-5


def _solve_gpu(ph, B, generic=False, fused=True, **kw):
    from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc
    c = quadrotor_lmpc(ph, device=0)
    c.debug_force_generic(generic)
    c.debug_use_fused(2 if fused == "group" else int(bool(fused)))
    x0, u0, yref = quadrotor_batch(B)
    r = c.optimizeBatch(x0, u0, yref=yref, want_active=True, **kw)
    import torch
    torch.cuda.synchronize()
    return c, (x0, u0, yref), r


def test_reference_known_answer_n10():
    """reference test/LMPC/test_common.cpp:89-237: cmd ~ [-0.9916, 1.74839, -0.9916, 1.74839] (rel 1e-4)"""
    c, _, r = _solve_gpu(10, 1)
    cmd = r.cmd[0].cpu().numpy()
    expect = np.array([-0.9916, 1.74839, -0.9916, 1.74839])
    assert np.linalg.norm(cmd - expect) <= 1e-4 * min(np.linalg.norm(cmd), np.linalg.norm(expect))
    assert int(r.status[0]) == 0 and int(r.solver_status[0]) == 1 and int(r.is_feasible[0]) == 1
    assert abs(float(r.cost[0]) - (-40.983485979)) < 1e-6


@pytest.mark.parametrize("path", ["group", "fused", "mfma-assemble", "generic-assemble"])
@pytest.mark.parametrize("ph,B", [(10, 64), (20, 256), (50, 32)])
def test_parity_with_oracle(ph, B, path):
    c, (x0, u0, yref), r = _solve_gpu(ph, B, generic=path == "generic-assemble", fused="group" if path == "group" else path == "fused")
    o = quadrotor_oracle(ph)
    ref = o.solve_batch_constref(x0, u0, yref, want_active=True)
    cmd = r.cmd.cpu().numpy(); cost = r.cost.cpu().numpy()
    st = r.status.cpu().numpy(); it = r.iterations.cpu().numpy()
    pol = ref["polished"] == 1
    assert pol.mean() > 0.95
    # statuses: everything the oracle solves must come back SUCCESS
    assert np.array_equal(st[ref["status"] == 0], np.zeros((ref["status"] == 0).sum(), dtype=st.dtype))
    scale = np.maximum(np.abs(ref["cmd"]).max(axis=1), 1e-12)
    err = np.abs(cmd - ref["cmd"]).max(axis=1) / scale
    assert err[pol].max() <= RTOL_CMD, (err[pol].max(), int(np.argmax(err * pol)))
    if (~pol).any():
        assert err[~pol].max() <= 5e-2
    cerr = np.abs(cost - ref["cost"]) / np.maximum(1.0, np.abs(ref["cost"]))
    assert cerr[pol].max() <= 1e-7
    # active sets on the inequality block, reference row numbering
    m = o.ncon
    lo = bits_to_rows(r.active_lower.cpu().numpy(), m); up = bits_to_rows(r.active_upper.cpu().numpy(), m)
    bad = 0
    for b in range(B):
        if not pol[b]:
            continue
        rl = np.nonzero(ref["active_lower"][b][o.neq:])[0] + o.neq
        ru = np.nonzero(ref["active_upper"][b][o.neq:])[0] + o.neq
        if not (np.array_equal(lo[b], rl) and np.array_equal(up[b], ru)):
            bad += 1
    assert bad == 0
    assert it.max() <= 250


# ---------------------------------------------------------------------------------------------
# edge cases and the rest of the LMPC surface
# ---------------------------------------------------------------------------------------------
from helpers import OracleFrontEnd, configure_quadrotor, configure_random, random_lmpc_spec, rel_err  # noqa: E402


@pytest.mark.parametrize("B", [0, 1, 3, 17, 65])
def test_ragged_batch_sizes(B):
    """empty and ragged batches: partial workgroups and partial MFMA tiles"""
    from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc
    import torch
    c = quadrotor_lmpc(10, device=0)
    x0, u0, yref = quadrotor_batch(max(B, 1))
    x0, u0, yref = x0[:B], u0[:B], yref[:B]
    r = c.optimizeBatch(x0, u0, yref=yref)
    torch.cuda.synchronize()
    assert tuple(r.cmd.shape) == (B, 4)
    if B:
        ref = quadrotor_oracle(10).solve_batch_constref(x0, u0, yref)
        pol = ref["polished"] == 1
        err = np.abs(r.cmd.cpu().numpy() - ref["cmd"]).max(axis=1) / np.maximum(np.abs(ref["cmd"]).max(axis=1), 1e-12)
        assert err[pol].max() <= RTOL_CMD


def test_single_optimize_matches_reference_interface():
    """IMPC::optimize / getLastResult / getOptimalSequence (IMPC.hpp:149-190) through the batch path"""
    from libmpc_amd.workloads import quadrotor_lmpc
    c = quadrotor_lmpc(10, device=0)
    res = c.optimize(np.zeros(12), np.zeros(4))
    f = configure_quadrotor(OracleFrontEnd(12, 4, 4, 12, 10, 10), 10)
    f.setOptimizerParameters(maximum_iteration=250)
    ref = f.optimize(np.zeros(12), np.zeros(4))
    assert rel_err(res.cmd, ref["cmd"]) <= RTOL_CMD and res.status == 0 and res.is_feasible
    assert c.getLastResult() is res
    seq = c.getOptimalSequence()
    assert seq.state.shape == (11, 12) and seq.input.shape == (11, 4) and seq.output.shape == (11, 12)
    assert np.allclose(seq.state, ref["state"], rtol=1e-6, atol=1e-8)
    assert np.allclose(seq.input, ref["input"], rtol=1e-6, atol=1e-8)
    assert np.allclose(seq.output, ref["output"], rtol=1e-6, atol=1e-8)
    assert np.array_equal(seq.state[0], np.zeros(12))       # row 0 = initial condition (CHANGELOG.md:51)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_full_feature_controller_parity(seed):
    """disturbances, per-step weights, sliced bounds, scalar constraint, move blocking (ch < ph),
    shared per-step references: generic assemble path, every output compared with the oracle"""
    from libmpc_amd import LMPC, LParameters
    import torch
    spec = random_lmpc_spec(seed)
    nx, nu, ndu, ny, ph, ch = spec["dims"]
    c = configure_random(LMPC(*spec["dims"], device=0), spec)
    c.setOptimizerParameters(LParameters(maximum_iteration=2000))
    f = configure_random(OracleFrontEnd(*spec["dims"]), spec)
    f.setOptimizerParameters(maximum_iteration=2000)
    r = np.random.default_rng(100 + seed)
    B = 24
    x0 = r.uniform(-1, 1, size=(B, nx)); x0[:, 0] *= 0.5
    u0 = r.uniform(-0.5, 0.5, size=(B, nu))
    out = c.optimizeBatch(x0, u0, want_active=True, want_sequence=True)
    torch.cuda.synchronize()
    m = f.o.ncon
    lo = bits_to_rows(out.active_lower.cpu().numpy(), m); up = bits_to_rows(out.active_upper.cpu().numpy(), m)
    checked = 0
    for b in range(B):
        ref = f.optimize(x0[b], u0[b])
        if ref["polished"] != 1:
            continue
        checked += 1
        assert int(out.status[b]) == 0
        assert rel_err(out.cmd[b].cpu().numpy(), ref["cmd"]) <= RTOL_CMD, b
        assert abs(float(out.cost[b]) - ref["cost"]) <= 1e-6 * max(1.0, abs(ref["cost"]))
        assert np.allclose(out.seq_state[b].cpu().numpy(), ref["state"], rtol=1e-5, atol=1e-7)
        assert np.allclose(out.seq_input[b].cpu().numpy(), ref["input"], rtol=1e-5, atol=1e-7)
        assert np.allclose(out.seq_output[b].cpu().numpy(), ref["output"], rtol=1e-5, atol=1e-7)
        rl = np.nonzero(ref["active_lower"][f.o.neq:])[0] + f.o.neq
        ru = np.nonzero(ref["active_upper"][f.o.neq:])[0] + f.o.neq
        # move blocking pins several identical input rows to one variable: compare as sets of
        # (bound side, condensed variable) there, exact rows elsewhere
        # ... and the delta-u rows pinned to zero past the control horizon are equalities (always
        # active, like the dynamics rows; the condensed QP has eliminated them): not reported
        na = nx + nu
        du0 = 2 * f.o.neq + (ph + 1) * ny
        nonu = lambda rows: [x for x in rows if not (x < 2 * f.o.neq and (x - f.o.neq) % na >= nx and (x - f.o.neq) // na > ch)
                             and not (du0 <= x < du0 + ph * nu)]
        assert nonu(lo[b]) == nonu(rl) and nonu(up[b]) == nonu(ru), (b, lo[b], rl, up[b], ru)
    assert checked >= B // 2


def test_per_step_references_and_per_instance_disturbances():
    """[B x ph x n] references and exogenous inputs: the reference's setReferences(matrix) /
    setExogenousInputs(matrix) per controller instance"""
    from libmpc_amd import LMPC, LParameters
    import torch
    spec = random_lmpc_spec(7)
    nx, nu, ndu, ny, ph, ch = spec["dims"]
    c = configure_random(LMPC(*spec["dims"], device=0), spec)
    c.setOptimizerParameters(LParameters(maximum_iteration=2000))
    f = configure_random(OracleFrontEnd(*spec["dims"]), spec)
    f.setOptimizerParameters(maximum_iteration=2000)
    r = np.random.default_rng(5)
    B = 12
    x0 = r.uniform(-0.5, 0.5, size=(B, nx)); u0 = r.uniform(-0.3, 0.3, size=(B, nu))
    yref = r.normal(size=(B, ph, ny)); uref = 0.05 * r.normal(size=(B, ph, nu))
    duref = 0.01 * r.normal(size=(B, ph, nu)); dmeas = 0.2 * r.normal(size=(B, ph, ndu))
    out = c.optimizeBatch(x0, u0, yref=yref, uref=uref, duref=duref, dmeas=dmeas)
    torch.cuda.synchronize()
    n = 0
    for b in range(B):
        ref = f.optimize(x0[b], u0[b], yRef=yref[b].T, uRef=uref[b].T, duRef=duref[b].T, dMeas=dmeas[b].T)
        if ref["polished"] == 1:
            n += 1
            assert rel_err(out.cmd[b].cpu().numpy(), ref["cmd"]) <= RTOL_CMD, b
            assert abs(float(out.cost[b]) - ref["cost"]) <= 1e-6 * max(1.0, abs(ref["cost"]))
    assert n >= B // 2


def test_reference_scalar_constraint_test():
    """test/LMPC/test_constraints.cpp:95-167 on the GPU path.  Its step-0 scalar row is violated by
    x0 itself; the reference still returns a usable sequence (see include/mpcx.h on infeasibility)."""
    import json, os
    import scipy.linalg as sla
    from libmpc_amd import LMPC, LParameters
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))["scalar_constraint_property"]
    nx, nu, ph = 2, 1, 5
    Mx = sla.expm(np.block([[np.array(g["A_continuous"], float), np.array(g["B_continuous"], float)], [np.zeros((nu, nx + nu))]]) * g["Ts"])
    c = LMPC(nx, nu, 0, 2, ph, ph, device=0)
    assert c.setStateSpaceModel(Mx[:nx, :nx], Mx[:nx, nx:], np.eye(2))
    assert c.setObjectiveWeights(g["OutputW"], g["InputW"], g["DeltaInputW"], (-1, -1))
    assert c.setScalarConstraint(g["smin"], g["smax"], np.ones(nx), np.ones(nu), (-1, -1))
    assert c.setReferences(np.zeros((2, ph)), np.zeros((nu, ph)), np.zeros((nu, ph)))
    c.setOptimizerParameters(LParameters(maximum_iteration=g["maximum_iteration"]))
    res = c.optimize(np.array(g["x0"]), np.array(g["u0"]))
    seq = c.getOptimalSequence()
    assert res.status == 1 and res.is_feasible            # MAX_ITERATION, as the reference reports it
    for i in range(ph):
        s = np.ones(nu) @ seq.input[i] + np.ones(nx) @ seq.state[i]
        assert s <= g["smax"] + g["tol_upper"] and s >= g["smin"] - g["tol_lower"]


def test_infeasible_instances_both_modes():
    from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc
    import torch
    x0, u0, yref = quadrotor_batch(8)
    x0[3, 0] = 1.0                                  # roll outside +-pi/6 at step 0
    u0[5, 1] = 5.0                                  # lastU outside the input box (quirk 2)
    c = quadrotor_lmpc(10, device=0)
    r = c.optimizeBatch(x0, u0, yref=yref); torch.cuda.synchronize()
    st = r.status.cpu().numpy(); cmd = r.cmd.cpu().numpy()
    assert st[3] == 1 and st[5] == 1 and np.isfinite(cmd).all() and (r.is_feasible.cpu().numpy() == 1).all()
    assert (np.delete(st, [3, 5]) == 0).all()
    ref = quadrotor_oracle(10).solve_batch_constref(x0, u0, yref)       # faithful: MAX_ITER_REACHED
    assert ref["status"][3] == 1 and ref["status"][5] == 1
    # the reference's answer there is an unconverged ADMM iterate: agreement is loose by nature.
    # 5: only the step-0 row on lastU is violated, the rest of the QP is solvable; 3: the roll
    # bound cannot be met anywhere along the horizon, any returned iterate is a compromise
    assert np.abs(cmd[5] - ref["cmd"][5]).max() < 0.05
    assert cmd[3].min() >= 9.6 - 10.5916 - 1e-9 and cmd[3].max() <= 13 - 10.5916 + 1e-9
    c.setStrictInfeasibility(True)
    r = c.optimizeBatch(x0, u0, yref=yref); torch.cuda.synchronize()
    st = r.status.cpu().numpy(); cmd = r.cmd.cpu().numpy()
    assert st[3] == 2 and st[5] == 2 and np.isnan(cmd[3]).all() and np.isnan(cmd[5]).all()
    assert (r.is_feasible.cpu().numpy()[[3, 5]] == 0).all() and float(r.cost[3]) == 1e30


def test_admm_only_mode():
    """polish = false: plain ADMM to OSQP's eps (1e-4); agreement with the polished oracle is at
    ADMM accuracy, status still SUCCESS"""
    from libmpc_amd import LParameters
    from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc
    import torch
    c = quadrotor_lmpc(10, device=0)
    c.setOptimizerParameters(LParameters(maximum_iteration=1000, polish=0))
    x0, u0, yref = quadrotor_batch(32)
    r = c.optimizeBatch(x0, u0, yref=yref); torch.cuda.synchronize()
    ref = quadrotor_oracle(10).solve_batch_constref(x0, u0, yref)
    it = r.iterations.cpu().numpy()
    assert (r.status.cpu().numpy() == 0).all() and it.min() >= 10 and it.max() <= 1000
    err = np.abs(r.cmd.cpu().numpy() - ref["cmd"]).max(axis=1) / np.maximum(np.abs(ref["cmd"]).max(axis=1), 1e-12)
    assert err.max() < 2e-2


def test_full_size_properties():
    """BASELINE config 2 at full size (N=20, B=4096): size-independent properties -- every instance
    solved, inputs inside their box, bit-identical across launches, instance 0 = the pinned answer,
    golden fixture for the first 64 instances"""
    import os
    from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc
    import torch
    c = quadrotor_lmpc(20, device=0)
    x0, u0, yref = quadrotor_batch(4096)
    a = c.optimizeBatch(x0, u0, yref=yref, want_sequence=True); torch.cuda.synchronize()
    b = c.optimizeBatch(x0, u0, yref=yref); torch.cuda.synchronize()
    assert torch.equal(a.cmd, b.cmd) and torch.equal(a.cost, b.cost)          # deterministic
    assert (a.status == 0).all() and (a.is_feasible == 1).all()
    u = a.seq_input.cpu().numpy()
    assert u.min() >= 9.6 - 10.5916 - 1e-7 and u.max() <= 13 - 10.5916 + 1e-7
    x = a.seq_state.cpu().numpy()
    assert np.abs(x[:, 1:, :2]).max() <= np.pi / 6 + 1e-7 and x[:, 1:, 5].min() >= -1 - 1e-7
    assert abs(float(a.cmd[0, 1]) - 1.7324892036) < 1e-9
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "quadrotor_oracle_n20.npz"))
    pol = g["polished"] == 1
    err = np.abs(a.cmd[:64].cpu().numpy() - g["cmd"]).max(axis=1) / np.maximum(np.abs(g["cmd"]).max(axis=1), 1e-12)
    assert err[pol].max() <= RTOL_CMD


def test_config2_whole_batch_against_oracle():
    """BASELINE config 2, every one of the 4096 instances against the C oracle (thread pool over the host cores): u* to 1e-5,
    cost to 1e-7 and the active set bit for bit wherever the oracle's polish succeeded; the count of the others is asserted
    (there the reference returns an eps = 1e-4 ADMM iterate, the GPU the exact optimum: compared to 5e-2)"""
    import torch
    from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc
    from helpers import assert_matches_oracle, oracle_batch_parallel
    B = 4096
    x0, u0, yref = quadrotor_batch(B)
    ref = oracle_batch_parallel(20, x0, u0, yref, want_active=True)
    o = quadrotor_oracle(20)
    for mode in (None, 0):                       # the default path at this batch (one workgroup per sixteen instances), and two kernels
        c = quadrotor_lmpc(20, device=0)
        c.debug_use_fused(mode)
        r = c.optimizeBatch(x0, u0, yref=yref, want_active=True); torch.cuda.synchronize()
        unpolished = assert_matches_oracle(r, ref, o.neq, o.ncon)
        assert unpolished <= 64, unpolished          # 29 of 4096 when this was written
        assert int(r.iterations.max()) == 0          # nothing needed the ADMM fallback
    print("config 2: 4096 instances compared, oracle polish failed on %d" % unpolished)


def test_full_size_properties_config4_shard():
    """BASELINE config 4's per-GPU shard (N=50, B=32768), on both forms -- the default (assemble, solve and cost kernels with the record in
    the workspace) and the in-workgroup form on request (lmpc_solve_group<2, 2>: eight instances per workgroup, two-chunk MFMA assemble,
    the solve and the cost from LDS; debug_use_fused(2)): every instance solved and feasible, inputs inside their box, bit-identical
    across launches, the first 2048 instances against the oracle; the two forms within 1e-9 of each other"""
    import torch
    from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc
    from helpers import assert_matches_oracle, oracle_batch_parallel
    B, n = 32768, 2048
    x0, u0, yref = quadrotor_batch(B)
    ref = oracle_batch_parallel(50, x0[:n], u0[:n], yref[:n], want_active=True)
    oq = quadrotor_oracle(50)
    cmds = {}
    for mode in (None, 2):
        c = quadrotor_lmpc(50, device=0)
        c.debug_use_fused(mode)
        a = c.optimizeBatch(x0, u0, yref=yref); torch.cuda.synchronize()
        b = c.optimizeBatch(x0, u0, yref=yref); torch.cuda.synchronize()
        assert torch.equal(a.cmd, b.cmd) and torch.equal(a.cost, b.cost)
        assert (a.status == 0).all() and (a.is_feasible == 1).all()
        u = a.cmd.cpu().numpy()
        assert u.min() >= 9.6 - 10.5916 - 1e-7 and u.max() <= 13 - 10.5916 + 1e-7
        # the first 2048 instances against the oracle (thread pool), active sets included
        r = c.optimizeBatch(x0[:n], u0[:n], yref=yref[:n], want_active=True); torch.cuda.synchronize()
        assert torch.equal(r.cmd, a.cmd[:n])                  # a prefix of the batch solved alone: the same bits
        unpolished = assert_matches_oracle(r, ref, oq.neq, oq.ncon)
        assert unpolished <= n // 20, unpolished
        cmds[mode] = (a.cmd.clone(), a.cost.clone())
    assert (cmds[None][0] - cmds[2][0]).abs().max().item() <= 1e-9
    assert ((cmds[None][1] - cmds[2][1]).abs() / cmds[None][1].abs().clamp(min=1.0)).max().item() <= 1e-9


def test_admm_fallback_is_reached_and_lands_on_the_oracle():
    """The polish-only kernel capped at one round: every instance whose first working set does not verify goes through
    lmpc_solve_admm -- ADMM iterations on the condensed QP, OSQP's active-set guess, polish again -- and must land on the same
    point as the oracle (u* 1e-5, cost 1e-7, identical active sets)."""
    import ctypes as C
    import torch
    from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc
    from helpers import assert_matches_oracle
    B = 256
    x0, u0, yref = quadrotor_batch(B)
    o = quadrotor_oracle(20)
    ref = o.solve_batch_constref(x0, u0, yref, want_active=True)
    c = quadrotor_lmpc(20, device=0)
    c.debug_use_fused(0)
    from libmpc_amd._capi import check
    check(c._lib.mpcx_lmpc_debug_set_rounds(c._h, 1, 10))
    r = c.optimizeBatch(x0, u0, yref=yref, want_active=True); torch.cuda.synchronize()
    it = r.iterations.cpu().numpy()
    assert (it > 0).mean() > 0.7, (it > 0).mean()          # most instances need more than one round: they took the ADMM path
    assert it.max() <= 250
    assert_matches_oracle(r, ref, o.neq, o.ncon)
    # and the same instances without the cap: identical results up to round-off, no ADMM iteration
    c2 = quadrotor_lmpc(20, device=0)
    r2 = c2.optimizeBatch(x0, u0, yref=yref, want_active=True); torch.cuda.synchronize()
    assert int(r2.iterations.max()) == 0
    np.testing.assert_allclose(r.cmd.cpu().numpy(), r2.cmd.cpu().numpy(), rtol=1e-7, atol=1e-9)
    assert torch.equal(r.active_lower, r2.active_lower) and torch.equal(r.active_upper, r2.active_upper)


def test_one_handle_large_then_small_batch_every_path():
    """One controller, a batch of 16384 and then one of 4096 on the same handle, for every solve path (the persistent fused kernel
    keeps work counters in the handle: they must not leak into the next launch): the small batch equals a fresh controller's."""
    import torch
    from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc
    xb, ub, yb = quadrotor_batch(16384)
    xs, us, ys = quadrotor_batch(4096, first=20000)
    fresh = quadrotor_lmpc(20, device=0)
    fresh.debug_use_fused(0)
    want = fresh.optimizeBatch(xs, us, yref=ys); torch.cuda.synchronize()
    assert (want.status == 0).all()
    for mode in (None, 0, 1, 2):
        c = quadrotor_lmpc(20, device=0)
        c.debug_use_fused(mode)
        big = c.optimizeBatch(xb, ub, yref=yb); torch.cuda.synchronize()
        assert (big.status == 0).all()
        small = c.optimizeBatch(xs, us, yref=ys); torch.cuda.synchronize()
        assert torch.equal(small.status, want.status)
        np.testing.assert_allclose(small.cmd.cpu().numpy(), want.cmd.cpu().numpy(), rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(small.cost.cpu().numpy(), want.cost.cpu().numpy(), rtol=1e-8, atol=1e-6)      # (the composed maps of the mat-vec form round differently)


@pytest.mark.parametrize("ph", [20, 50])
def test_instances_dealt_to_the_wavefronts_are_solved_as_in_place(ph):
    """lmpc_solve_group deals the sixteen (N = 50: eight) instances of a workgroup to its wavefronts by how far their unconstrained optimum lies outside
    the bounds (the launch is its busiest SIMD's work); which wavefront solves an instance must not show: whole workgroups, a partial last
    one (kept in place) and a batch below one workgroup against the two-kernel path, where every instance is solved by the wavefront of its index --
    commands, costs, statuses, rounds and active sets bit for bit."""
    import torch
    from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc
    for B, first in ((5, 0), (16, 100), (31, 200), (1024 + 7, 300)):
        x0, u0, yref = quadrotor_batch(B, first=first)
        got = {}
        for mode in (2, 0):
            c = quadrotor_lmpc(ph, device=0)
            c.debug_use_fused(mode)
            r = c.optimizeBatch(x0, u0, yref=yref, want_active=True); torch.cuda.synchronize()
            assert (r.status == 0).all()
            got[mode] = r
        a, b = got[2], got[0]
        for name in ("cmd", "status", "solver_status", "polish_rounds", "active_count", "active_lower", "active_upper"):
            assert torch.equal(getattr(a, name), getattr(b, name)), (name, B)
        if ph == 20:
            assert torch.equal(a.cost, b.cost), B                # (N = 50: the group form takes the cost from its definition inside the solve, the other from lmpc_cost_mfma)
        else:
            np.testing.assert_allclose(a.cost.cpu().numpy(), b.cost.cpu().numpy(), rtol=1e-9)


def test_shards_equal_rows_of_the_unsharded_solve():
    """Multi-GPU readiness on one device: the batch of 8 x 512 instances solved as eight shards (what eight ranks would do,
    quadrotor_batch(B, first = r B)) gives, bit for bit, the rows of the unsharded solve."""
    import torch
    from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc
    Bs, R = 512, 8
    x0, u0, yref = quadrotor_batch(Bs * R)
    for mode in (None, 0):
        c = quadrotor_lmpc(20, device=0)
        c.debug_use_fused(mode)
        whole = c.optimizeBatch(x0, u0, yref=yref, want_active=True); torch.cuda.synchronize()
        for rk in range(R):
            xs, us, ys = quadrotor_batch(Bs, first=rk * Bs)
            if rk == 0:
                assert np.array_equal(xs, x0[:Bs])
            else:
                assert np.array_equal(xs, x0[rk * Bs:(rk + 1) * Bs]) and np.array_equal(ys, yref[rk * Bs:(rk + 1) * Bs])
            part = c.optimizeBatch(xs, us, yref=ys, want_active=True); torch.cuda.synchronize()
            sl = slice(rk * Bs, (rk + 1) * Bs)
            assert torch.equal(part.cmd, whole.cmd[sl]) and torch.equal(part.cost, whole.cost[sl]) and torch.equal(part.status, whole.status[sl])
            assert torch.equal(part.active_lower, whole.active_lower[sl]) and torch.equal(part.active_upper, whole.active_upper[sl])


def test_shards_across_the_form_threshold_equal_rows_of_the_unsharded_solve():
    """The library picks the kernel form by batch size (one workgroup per sixteen instances up to 4096 instances, assemble + solve as two
    kernels beyond).  A rank that is given a shard tells its handle the size of the whole batch (setTotalBatch, mpcx_lmpc_set_total_batch):
    the shard then takes the form the whole batch would, and its results are bit for bit the rows of the unsharded solve -- 2 x 4096 against
    8192 here, the two sizes on either side of the threshold.  Without the hint the two forms agree to 1e-9, not in every bit."""
    import torch
    from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc
    Bs, R = 4096, 2
    x0, u0, yref = quadrotor_batch(Bs * R)
    c = quadrotor_lmpc(20, device=0)
    whole = c.optimizeBatch(x0, u0, yref=yref, want_active=True); torch.cuda.synchronize()
    shard = quadrotor_lmpc(20, device=0)
    assert shard.setTotalBatch(Bs * R)
    for rk in range(R):
        sl = slice(rk * Bs, (rk + 1) * Bs)
        part = shard.optimizeBatch(x0[sl], u0[sl], yref=yref[sl], want_active=True); torch.cuda.synchronize()
        assert torch.equal(part.cmd, whole.cmd[sl]) and torch.equal(part.cost, whole.cost[sl]) and torch.equal(part.status, whole.status[sl])
        assert torch.equal(part.active_lower, whole.active_lower[sl]) and torch.equal(part.active_upper, whole.active_upper[sl])
    plain = quadrotor_lmpc(20, device=0)                       # no hint: the in-workgroup form
    part = plain.optimizeBatch(x0[:Bs], u0[:Bs], yref=yref[:Bs]); torch.cuda.synchronize()
    ok = (part.status == 0) & (whole.status[:Bs] == 0)
    assert ((part.cmd - whole.cmd[:Bs]).abs().max(dim=1).values[ok] <= 1e-9 * whole.cmd[:Bs].abs().max(dim=1).values[ok].clamp(min=1.0)).all()


def test_warm_start_carries_the_working_set():
    """f1: the previous tick's active set seeds the working set (LOptimizer.hpp:268-281 carries x, y): same results,
    an unchanged active set verifies in one round, a plant step needs fewer rounds than a cold start."""
    import torch
    from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc
    ph, B = 20, 512
    c = quadrotor_lmpc(ph, device=0)
    x0, u0, yref = quadrotor_batch(B)
    cold = c.optimizeBatch(x0, u0, yref=yref, want_active=True)
    again = c.optimizeBatch(x0, u0, yref=yref, warm=cold)
    torch.cuda.synchronize()
    ok = (cold.status == 0)
    assert torch.equal(again.status, cold.status)
    np.testing.assert_allclose(again.cmd.cpu().numpy(), cold.cmd.cpu().numpy(), rtol=1e-9, atol=1e-12)
    assert torch.equal(again.active_lower, cold.active_lower) and torch.equal(again.active_upper, cold.active_upper)
    assert (again.polish_rounds[ok] == 1).all()
    # one plant step x+ = A x + B (u_trim + cmd) for every instance, then the next tick warm and cold
    from libmpc_amd.workloads import quadrotor_matrices
    Ad, Bd, _ = quadrotor_matrices()
    x1 = torch.as_tensor(x0).cuda() @ torch.as_tensor(Ad).cuda().T + cold.cmd @ torch.as_tensor(Bd).cuda().T
    nxt_cold = c.optimizeBatch(x1, cold.cmd, yref=yref, want_active=True)
    nxt_warm = c.optimizeBatch(x1, cold.cmd, yref=yref, warm=cold, warm_shift=True)
    torch.cuda.synchronize()
    assert torch.equal(nxt_warm.status, nxt_cold.status)
    ok = (nxt_cold.status == 0)
    err = (nxt_warm.cmd - nxt_cold.cmd).abs().max(dim=1).values / nxt_cold.cmd.abs().max(dim=1).values.clamp_min(1e-12)
    assert err[ok].max().item() <= 1e-8
    assert torch.equal(nxt_warm.active_lower[ok], nxt_cold.active_lower[ok])
    assert nxt_warm.polish_rounds[ok].float().mean().item() < nxt_cold.polish_rounds[ok].float().mean().item()
    print("rounds cold %.2f warm %.2f" % (nxt_cold.polish_rounds[ok].float().mean().item(), nxt_warm.polish_rounds[ok].float().mean().item()))


def test_single_instance_front_end_warm_start_and_stats():
    """pybind surface (python/pybind_export.cpp:93-123): optimize() with enable_warm_start, getExecutionStats,
    get/setSolverWarmStart* -- the closed loop of examples/quadrotor_ex.cpp for a few ticks, warm vs cold"""
    from libmpc_amd import LParameters
    from libmpc_amd.workloads import quadrotor_lmpc, quadrotor_matrices
    Ad, Bd, _ = quadrotor_matrices()
    cold = quadrotor_lmpc(10, device=0)
    warm = quadrotor_lmpc(10, device=0)
    warm.setOptimizerParameters(LParameters(maximum_iteration=250, enable_warm_start=1))
    x = np.zeros(12); u = np.zeros(4)
    for k in range(6):
        rc = cold.optimize(x, u); rw = warm.optimize(x, u)
        assert rc.status == rw.status == 0
        np.testing.assert_allclose(rw.cmd, rc.cmd, rtol=1e-9, atol=1e-12)
        x = Ad @ x + Bd @ rc.cmd; u = rc.cmd
    st = warm.getExecutionStats()
    assert st.numberOfSolutions == 6 and st.minSolutionTime <= st.averageSolutionTime <= st.maxSolutionTime
    assert st.solutionsStates == {0: 6}
    y = warm.getSolverWarmStartDual(); z = warm.getSolverWarmStartPrimal()
    i = warm.info()
    assert y.shape == (i["m_ref"],) and z.shape == (i["n_ref"],) and set(np.unique(y)) <= {-1.0, 0.0, 1.0}
    other = quadrotor_lmpc(10, device=0)
    other.setOptimizerParameters(LParameters(maximum_iteration=250, enable_warm_start=1))
    other.setSolverWarmStart(z, y)
    ro = other.optimize(x, u); rc = cold.optimize(x, u)
    np.testing.assert_allclose(ro.cmd, rc.cmd, rtol=1e-9, atol=1e-12)
    warm.resetStats()
    assert warm.getExecutionStats().numberOfSolutions == 0


def test_reference_refresh_gives_the_results_of_a_fresh_controller():
    """setReferences / setExogenousInputs between solves: the in-place refresh of the reference-dependent device data must give
    exactly what a controller built with those references from the start gives (shared and per-instance reference paths)"""
    import torch
    from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc
    B = 96
    x0, u0, yref_b = quadrotor_batch(B)
    a = quadrotor_lmpc(20, device=0)
    r0 = a.optimizeBatch(x0, u0)                          # first solve with the example's references
    torch.cuda.synchronize()
    yr = np.zeros(12); yr[2] = 0.6; yr[9] = 0.05
    ur = np.full(4, 0.02)
    assert a.setReferences(yr, ur, np.zeros(4), (0, 20))
    b = quadrotor_lmpc(20, device=0)
    assert b.setReferences(yr, ur, np.zeros(4), (0, 20))
    for kw in (dict(), dict(yref=yref_b)):
        ra = a.optimizeBatch(x0, u0, **kw); rb = b.optimizeBatch(x0, u0, **kw)
        torch.cuda.synchronize()
        # equal up to round-off: the kept linear columns of the assemble maps were tabulated as differences around the OLD
        # references, those of the fresh controller around the new ones -- the same numbers to the last bit or two
        assert torch.equal(ra.status, rb.status)
        np.testing.assert_allclose(ra.cmd.cpu().numpy(), rb.cmd.cpu().numpy(), rtol=1e-8, atol=1e-10)      # measured: 6e-12 absolute
        # the optimal cost is c0 + (a term of the solution): c0 carries ref' W ref, far larger than the cost, so the last bits of
        # the kept columns show up as ~1e-8 of the cost (measured 2e-8)
        np.testing.assert_allclose(ra.cost.cpu().numpy(), rb.cost.cpu().numpy(), rtol=2e-7, atol=1e-9)
    assert not torch.equal(r0.cmd, ra.cmd)


def test_a_step_replayed_as_a_hip_graph_gives_the_plain_launch_s_results():
    """mpcx_lmpc_graph_*: the launches of one step captured once and replayed; new inputs written into the descriptor's
    tensors in place are what the replay solves"""
    import torch
    from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc
    c = quadrotor_lmpc(20, device=0)
    B = 512
    x0, u0, yref = quadrotor_batch(2 * B)
    batch, res, keep = c.make_batch(x0[:B], u0[:B], yref=yref[:B])
    side = torch.cuda.Stream(device=0)
    g = c.make_graph(batch, side)
    c.launch_graph(g, side); side.synchronize()
    plain = c.optimizeBatch(x0[:B], u0[:B], yref=yref[:B]); torch.cuda.synchronize()
    assert torch.equal(res.cmd, plain.cmd) and torch.equal(res.cost, plain.cost) and torch.equal(res.status, plain.status)
    # the next tick's data, in place
    xk, uk, yk = [t for t in keep if t is not None][:3]
    xk.copy_(torch.from_numpy(x0[B:]).to(xk.device)); uk.copy_(torch.from_numpy(u0[B:]).to(uk.device)); yk.copy_(torch.from_numpy(yref[B:]).to(yk.device))
    torch.cuda.synchronize()
    for _ in range(3):
        c.launch_graph(g, side)
    side.synchronize()
    plain2 = c.optimizeBatch(x0[B:], u0[B:], yref=yref[B:]); torch.cuda.synchronize()
    assert torch.equal(res.cmd, plain2.cmd) and torch.equal(res.cost, plain2.cost)
    assert not torch.equal(plain.cmd, plain2.cmd)
    c.destroy_graph(g)


def test_overlapped_gather_as_one_hip_graph_per_parity_gathers_every_step():
    """libmpc_amd.distributed.GraphedOverlap: {solve of step k || RCCL all-gather of step k-1} captured as one HIP graph per buffer parity
    (one rank here: the collective gathers the block onto itself).  Inputs change between launches (the descriptors' pointers are baked in,
    what they point to is not): after every launch the gathered controls of the previous step are that step's results, and the last step's
    arrive with flush()."""
    import torch
    from libmpc_amd.distributed import ControlGather, GraphedOverlap, OverlappedGather
    from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc
    B = 256
    c = quadrotor_lmpc(20, device=0)
    x0, u0, yref = quadrotor_batch(B)
    b0, r0, keep0 = c.make_batch(x0, u0, yref=yref)
    b1, r1, keep1 = c.make_batch(x0, u0, yref=yref)
    pair, res = [b0, b1], [r0, r1]
    g = ControlGather(0, 0, 1)
    og = OverlappedGather(B, 4, torch.device("cuda:0"), gather=g, solve_stream=torch.cuda.current_stream(0))
    og.cmd = [r0.cmd, r1.cmd]
    go = GraphedOverlap(og, lambda i, s: c.launch(pair[i], s))
    ref = c.optimizeBatch(x0, u0, yref=yref).cmd.clone(); torch.cuda.synchronize()
    prev = None
    for k in range(5):
        go.step(); go.stream.synchronize()
        i = (go.k - 1) % 2
        assert torch.equal(res[i].cmd, ref)                     # the same inputs every step: the same controls
        if prev is not None:
            assert torch.equal(og.all[prev], ref)               # the previous step's controls were gathered by this graph
        prev = i
    assert torch.equal(go.flush(), ref)
