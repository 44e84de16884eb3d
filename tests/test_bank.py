"""LMPCBank: a mixed batch over a few different controllers (heterogeneity by grouping, libmpc_amd/bank.py)."""
import numpy as np
import pytest


def test_group_by_model_is_a_stable_partition():
    from libmpc_amd.bank import group_by_model
    model = np.array([2, 0, 1, 2, 2, 0, 1, 0])
    order, off = group_by_model(model, 4)                 # controller 3 serves nobody
    assert off.tolist() == [0, 3, 5, 8, 8]
    assert order[off[0]:off[1]].tolist() == [1, 5, 7]     # original relative order inside a group
    assert order[off[1]:off[2]].tolist() == [2, 6]
    assert order[off[2]:off[3]].tolist() == [0, 3, 4]
    assert sorted(order.tolist()) == list(range(8))
    with pytest.raises(ValueError):
        group_by_model([0, 4], 4)
    o, f = group_by_model([], 2)
    assert o.size == 0 and f.tolist() == [0, 0, 0]


@pytest.mark.gpu
def test_bank_results_are_those_of_each_instance_s_own_controller():
    """three quadrotor variants (other weights, other input bounds, other horizon of control moves) in one shuffled batch:
    every instance comes back with exactly what its own controller returns for it"""
    import torch
    from libmpc_amd import LMPCBank
    from libmpc_amd.workloads import quadrotor_batch, quadrotor_lmpc
    a = quadrotor_lmpc(20)
    b = quadrotor_lmpc(20)
    b.setObjectiveWeights([0, 0, 30, 10, 10, 10, 0, 0, 0, 5, 5, 5], [0.3] * 4, [0.1] * 4, (0, 20))
    c = quadrotor_lmpc(20)
    c.setInputBounds([-0.6] * 4, [1.2] * 4, (0, 20))
    bank = LMPCBank([a, b, c])
    B = 700
    x0, u0, yref = quadrotor_batch(B)
    rng = np.random.default_rng(3)
    model = rng.integers(0, 3, size=B)
    r = bank.optimizeBatch(x0, u0, model, yref=yref, want_sequence=True)
    torch.cuda.synchronize()
    for k, ctl in enumerate((a, b, c)):
        idx = np.nonzero(model == k)[0]
        rk = ctl.optimizeBatch(x0[idx], u0[idx], yref=yref[idx], want_sequence=True)
        torch.cuda.synchronize()
        sel = torch.from_numpy(idx).cuda()
        assert torch.equal(r.cmd[sel], rk.cmd) and torch.equal(r.cost[sel], rk.cost)
        assert torch.equal(r.status[sel], rk.status) and torch.equal(r.solver_status[sel], rk.solver_status)
        assert torch.equal(r.seq_state[sel], rk.seq_state)
    assert not torch.equal(r.cmd[torch.from_numpy(np.nonzero(model == 0)[0][:8]).cuda()],
                           r.cmd[torch.from_numpy(np.nonzero(model == 1)[0][:8]).cuda()])


def test_hetero_bank_refuses_to_exist_without_a_device():
    """mpcx_lmpc_hetero_create: the bank lives in HBM; host-only controllers configure it, nothing solves on the CPU"""
    import pytest
    import torch
    from libmpc_amd import LMPCHetero, MpcxError
    from libmpc_amd.workloads import quadrotor_lmpc
    if torch.cuda.is_available():
        pytest.skip("a device is present: covered by tests/test_lmpc_hetero.py")
    a = quadrotor_lmpc(10, device=-1)
    with pytest.raises(MpcxError, match="no CPU fallback"):
        LMPCHetero([a, a], device=0)
    with pytest.raises(ValueError):
        LMPCHetero([], device=0)
