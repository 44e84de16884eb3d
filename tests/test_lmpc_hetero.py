"""GPU parity tests of heterogeneous LMPC batches (mpcx_lmpc_hetero_*): every instance its own controller -- own A, B, C,
weights, bounds, references -- against the CPU oracle's front-end, one oracle controller per instance, exactly as the
reference would hold one mpc::LMPC<> object per problem (LMPC.hpp:751)."""
import numpy as np
import pytest

from helpers import OracleFrontEnd, bits_to_rows, configure_quadrotor, configure_random, quadrotor_oracle, random_lmpc_spec

pytestmark = pytest.mark.gpu


def _random_family(K, seed0=100):
    """K full-feature controllers (disturbances, per-step weights, slices of bounds, scalar constraint, move blocking) with the
    same dimensions and the same pattern of finite bounds, everything else different"""
    specs = [random_lmpc_spec(seed0 + k) for k in range(K)]
    return specs


def test_every_instance_its_own_random_controller_matches_the_oracle():
    import torch
    from libmpc_amd import LMPC, LMPCHetero, LParameters
    K = 48
    specs = _random_family(K)
    ctrls, oracles = [], []
    for sp in specs:
        c = configure_random(LMPC(*sp["dims"], device=-1), sp)
        c.setOptimizerParameters(LParameters(maximum_iteration=2000))
        ctrls.append(c)
        o = configure_random(OracleFrontEnd(*sp["dims"]), sp)
        o.setOptimizerParameters(maximum_iteration=2000)
        oracles.append(o)
    het = LMPCHetero(ctrls, device=0)
    assert het.count == K
    rng = np.random.default_rng(7)
    nx, nu = specs[0]["dims"][0], specs[0]["dims"][1]
    x0 = rng.uniform(-1, 1, size=(K, nx)); x0[:, 0] *= 0.5
    u0 = rng.uniform(-0.5, 0.5, size=(K, nu))
    r = het.optimizeBatch(x0, u0, want_active=True, want_sequence=True); torch.cuda.synchronize()
    cmd = r.cmd.cpu().numpy(); cost = r.cost.cpu().numpy(); st = r.status.cpu().numpy()
    checked = 0
    for k in range(K):
        ref = oracles[k].optimize(x0[k], u0[k])
        if ref["polished"] != 1:
            continue
        checked += 1
        assert st[k] == 0
        assert np.abs(cmd[k] - ref["cmd"]).max() <= 1e-5 * max(np.abs(ref["cmd"]).max(), 1e-12), (k, cmd[k], ref["cmd"])
        assert abs(cost[k] - ref["cost"]) <= 1e-6 * max(1.0, abs(ref["cost"]))
        assert np.allclose(r.seq_state[k].cpu().numpy(), ref["state"], rtol=1e-5, atol=1e-7)
        assert np.allclose(r.seq_input[k].cpu().numpy(), ref["input"], rtol=1e-5, atol=1e-7)
        assert np.allclose(r.seq_output[k].cpu().numpy(), ref["output"], rtol=1e-5, atol=1e-7)
    assert checked >= K * 3 // 4, checked
    # a bank is K controllers: every instance equals what its controller returns on its own through the shared-model path (whose
    # active sets tests/test_lmpc_gpu.py pins to the oracle's, move blocking included) -- commands, costs and active sets
    for k in range(K):
        c = configure_random(LMPC(*specs[k]["dims"], device=0), specs[k])
        c.setOptimizerParameters(LParameters(maximum_iteration=2000))
        one = c.optimizeBatch(x0[k:k + 1], u0[k:k + 1], want_active=True); torch.cuda.synchronize()
        np.testing.assert_allclose(one.cmd.cpu().numpy()[0], cmd[k], rtol=1e-9, atol=1e-11)
        assert int(one.status[0]) == st[k]
        assert torch.equal(one.active_lower[0], r.active_lower[k]) and torch.equal(one.active_upper[0], r.active_upper[k]), k


def test_quadrotor_variants_with_a_model_index():
    """a fleet: 64 quadrotor variants (scaled dynamics, own weights and input limits), 512 instances drawing their controller from
    an index; every instance against the C oracle of its own controller"""
    import torch
    from libmpc_amd import LMPC, LMPCHetero
    from libmpc_amd.workloads import quadrotor_batch, quadrotor_variant
    K, B, ph = 64, 512, 10
    ctrls = [quadrotor_variant(k, ph, device=-1) for k in range(K)]
    het = LMPCHetero(ctrls, device=0)
    x0, u0, yref = quadrotor_batch(B)
    model = (np.arange(B) * 7) % K
    r = het.optimizeBatch(x0, u0, model=model, yref=yref, want_active=True); torch.cuda.synchronize()
    cmd = r.cmd.cpu().numpy(); st = r.status.cpu().numpy()
    assert (st == 0).all()
    lo = bits_to_rows(r.active_lower.cpu().numpy(), het.m_ref); up = bits_to_rows(r.active_upper.cpu().numpy(), het.m_ref)
    worst = 0.0
    for k in range(K):
        idx = np.nonzero(model == k)[0]
        f = OracleFrontEnd(12, 4, 4, 12, ph, ph)
        quadrotor_variant(k, ph, into=f)
        from oracle.lmpc_oracle import default_params
        f.o.params = default_params(maximum_iteration=250)
        ref = f.o.solve_batch_constref(x0[idx], u0[idx], yref[idx], want_active=True)
        pol = ref["polished"] == 1
        err = np.abs(cmd[idx] - ref["cmd"]).max(axis=1) / np.maximum(np.abs(ref["cmd"]).max(axis=1), 1e-12)
        assert err[pol].max() <= 1e-5
        worst = max(worst, err[pol].max())
        for t, b in enumerate(idx):
            if pol[t]:
                rl = np.nonzero(ref["active_lower"][t][f.o.neq:])[0] + f.o.neq; ru = np.nonzero(ref["active_upper"][t][f.o.neq:])[0] + f.o.neq
                assert np.array_equal(lo[b], rl) and np.array_equal(up[b], ru)
    print("fleet of %d variants, %d instances: worst relative error of u* %.2e" % (K, B, worst))


def test_structure_mismatch_is_refused():
    from libmpc_amd import LMPC, LMPCHetero, MpcxError
    from libmpc_amd.workloads import quadrotor_variant
    a = quadrotor_variant(0, 10, device=-1)
    b = quadrotor_variant(1, 10, device=-1)
    b.setStateBounds([-1.0] * 12, [1.0] * 12, (0, 10))          # more finite bounds than controller 0: another row structure
    with pytest.raises(MpcxError):
        LMPCHetero([a, b], device=0)


@pytest.mark.parametrize("family", ["quadrotor", "random", "quadrotor50"])
def test_device_condensing_matches_the_host_set_up(family):
    """the bank's O(n^3) arrays -- Hessian, constraint rows, dual Hessian, ADMM matrix -- computed by lmpc_condense_models (MFMA)
    against LmpcController::condense on the host, array by array, and the solves of the two banks against each other"""
    import torch
    from libmpc_amd import LMPC, LMPCHetero, LParameters
    from libmpc_amd.workloads import quadrotor_batch, quadrotor_variant
    if family in ("quadrotor", "quadrotor50"):
        # (N = 50, config 4's shape: 200 condensed variables -- P and Q of the kernel in a scratch block per workgroup, not in LDS)
        K = 24 if family == "quadrotor" else 10
        ctrls = [quadrotor_variant(k, 20 if family == "quadrotor" else 50, device=-1) for k in range(K)]
        x0, u0, yref = quadrotor_batch(K)
    else:
        K = 24
        ctrls = []
        for sp in _random_family(K):
            c = configure_random(LMPC(*sp["dims"], device=-1), sp)
            c.setOptimizerParameters(LParameters(maximum_iteration=2000))
            ctrls.append(c)
        rng = np.random.default_rng(3)
        x0 = rng.uniform(-1, 1, size=(K, ctrls[0].nx)); x0[:, 0] *= 0.5
        u0 = rng.uniform(-0.5, 0.5, size=(K, ctrls[0].nu)); yref = None
    host = LMPCHetero(ctrls, device=0, condense_on_host=True)
    dev = LMPCHetero(ctrls, device=0)
    assert host.debug_get(0, "flags")[1] == 0.0 and dev.debug_get(0, "flags")[1] == 1.0
    for k in (0, 1, K // 2, K - 1):
        # (the inverses are as good as the Hessian's conditioning lets them be: two orders looser at 200 variables than at 80)
        loose = 100.0 if family == "quadrotor50" else 1.0
        for name, tol in (("H", 1e-12), ("Gr", 1e-12), ("Gc", 1e-12), ("Y", 1e-9 * loose), ("rho_b", 1e-9 * loose), ("rho_g", 1e-9 * loose), ("Kinv", 1e-8 * loose)):
            a, b = host.debug_get(k, name), dev.debug_get(k, name)
            scale = max(np.abs(a).max(), 1e-300)
            assert np.abs(a - b).max() <= tol * scale, (k, name, np.abs(a - b).max() / scale)
        assert host.debug_get(k, "flags")[0] == dev.debug_get(k, "flags")[0]
    ra = host.optimizeBatch(x0, u0, yref=yref, want_active=True); rb = dev.optimizeBatch(x0, u0, yref=yref, want_active=True)
    torch.cuda.synchronize()
    assert torch.equal(ra.status, rb.status)
    np.testing.assert_allclose(ra.cmd.cpu().numpy(), rb.cmd.cpu().numpy(), rtol=1e-8 * loose, atol=1e-10 * loose)
    np.testing.assert_allclose(ra.cost.cpu().numpy(), rb.cost.cpu().numpy(), rtol=1e-8 * loose, atol=1e-8 * loose)
    assert torch.equal(ra.active_lower, rb.active_lower) and torch.equal(ra.active_upper, rb.active_upper)
    if family == "quadrotor50":
        # ... and the bank against one oracle controller per instance (the N = 50 bank: device condensing in its scratch-block form)
        from oracle.lmpc_oracle import default_params
        assert (rb.status == 0).all()
        for k in range(K):
            f = OracleFrontEnd(12, 4, 4, 12, 50, 50)
            quadrotor_variant(k, 50, into=f)
            f.o.params = default_params(maximum_iteration=250)
            ref = f.o.solve_batch_constref(x0[k:k + 1], u0[k:k + 1], yref[k:k + 1])
            if ref["status"][0] == 0 and ref["polished"][0] == 1:
                np.testing.assert_allclose(rb.cmd.cpu().numpy()[k], ref["cmd"][0], rtol=1e-5, atol=1e-7)
        print("N = 50 bank of %d controllers: the condensing kernel %.2f ms, the bank's set-up %.1f ms (on the host's cores: %.1f ms)"
              % (K, dev.debug_get(0, "flags")[2], dev.debug_get(0, "flags")[3], host.debug_get(0, "flags")[3]))


def _two_state_controller(b1, hessian_weight=1.0):
    """x = [p, q]: p driven by the input, q driven by the input only through b1; q carries a bound, so its rows of the condensed G
    vanish identically when b1 = 0"""
    from libmpc_amd import LMPC, LParameters
    c = LMPC(2, 1, 0, 2, 6, 6, device=-1)
    c.setStateSpaceModel(np.array([[0.9, 0.0], [0.0, 0.8]]), np.array([[0.5], [b1]]), np.eye(2))
    c.setObjectiveWeights([hessian_weight, hessian_weight], [0.1], [0.0], (0, 6))
    c.setStateBounds([-np.inf, -1.0], [np.inf, 1.0], (0, 6))
    c.setInputBounds([-1.0], [1.0], (0, 6))
    c.setReferences([0.5, 0.0], [0.0], [0.0], (0, 6))
    c.setOptimizerParameters(LParameters(maximum_iteration=500))
    return c


@pytest.mark.parametrize("on_host", [False, True])
def test_a_bank_whose_constraint_structure_differs_from_controller_0_is_refused_on_both_condensing_paths(on_host):
    """the split into feasibility-only and general rows is taken from controller 0; a controller in which a bounded state does not depend on
    the inputs has that row identically zero in G -- the host path reports it, and so does the device condensing (cond_status), instead of
    condensing a singular row into the dual Hessian"""
    from libmpc_amd import LMPCHetero
    good = [_two_state_controller(0.3), _two_state_controller(0.25)]
    het = LMPCHetero(good, device=0, condense_on_host=on_host)
    assert het.count == 2
    with pytest.raises(Exception) as ei:
        LMPCHetero([_two_state_controller(0.3), _two_state_controller(0.0)], device=0, condense_on_host=on_host)
    assert "differs from controller 0" in str(ei.value) or "controller 1" in str(ei.value), str(ei.value)


def _variant_oracle_chunk(job):
    """worker: instances lo..hi of the heterogeneous bench batch, each on the C oracle of its own quadrotor_variant"""
    lo, hi, ph, x0, u0, yref = job
    from libmpc_amd.workloads import quadrotor_variant
    from oracle.lmpc_oracle import default_params
    cmd = np.zeros((hi - lo, 4)); cost = np.zeros(hi - lo); pol = np.zeros(hi - lo, dtype=bool); act = []
    for t, k in enumerate(range(lo, hi)):
        f = OracleFrontEnd(12, 4, 4, 12, ph, ph)
        quadrotor_variant(k, ph, into=f)
        f.o.params = default_params(maximum_iteration=250)
        ref = f.o.solve_batch_constref(x0[t:t + 1], u0[t:t + 1], yref[t:t + 1], want_active=True)
        cmd[t] = ref["cmd"][0]; cost[t] = ref["cost"][0]; pol[t] = ref["polished"][0] == 1
        act.append((np.nonzero(ref["active_lower"][0][f.o.neq:])[0] + f.o.neq, np.nonzero(ref["active_upper"][0][f.o.neq:])[0] + f.o.neq))
    return cmd, cost, pol, act


def test_the_bench_batch_of_4096_distinct_controllers_matches_one_oracle_controller_per_instance():
    """`bench.py --workload lmpc-hetero` on the batch it is quoted at: 4096 instances, instance k on controller quadrotor_variant(k)
    (N = 20), every one against the C oracle configured as that variant (worker processes, one oracle controller per instance) --
    commands to 1e-5, costs to 1e-6, active sets bit for bit wherever the oracle polished"""
    import multiprocessing as mp
    import os
    import torch
    from libmpc_amd import LMPCHetero
    from libmpc_amd.workloads import quadrotor_batch, quadrotor_variant
    B, ph = 4096, 20
    x0, u0, yref = quadrotor_batch(B)
    workers = max(1, min(16, len(os.sched_getaffinity(0))))
    edges = np.linspace(0, B, 4 * workers + 1).astype(int)
    jobs = [(int(a), int(b), ph, x0[a:b], u0[a:b], yref[a:b]) for a, b in zip(edges[:-1], edges[1:])]
    with mp.get_context("fork").Pool(workers) as pool:            # (forked before the first GPU call of this test)
        parts = pool.map(_variant_oracle_chunk, jobs)
    ocmd = np.concatenate([p[0] for p in parts]); ocost = np.concatenate([p[1] for p in parts]); pol = np.concatenate([p[2] for p in parts])
    oact = [a for p in parts for a in p[3]]
    het = LMPCHetero([quadrotor_variant(k, ph, device=-1) for k in range(B)], device=0)
    r = het.optimizeBatch(x0, u0, yref=yref, want_active=True); torch.cuda.synchronize()
    cmd = r.cmd.cpu().numpy(); cost = r.cost.cpu().numpy(); st = r.status.cpu().numpy()
    assert (st[pol] == 0).all()
    err = np.abs(cmd - ocmd).max(axis=1) / np.maximum(np.abs(ocmd).max(axis=1), 1e-12)
    cerr = np.abs(cost - ocost) / np.maximum(1.0, np.abs(ocost))
    assert pol.sum() >= 0.95 * B, pol.sum()
    assert err[pol].max() <= 1e-5, (err[pol].max(), int(np.argmax(err * pol)))
    assert cerr[pol].max() <= 1e-6, cerr[pol].max()          # (the bank is condensed on the device: the tolerance of the 48-controller test above)
    if (~pol).any():
        assert err[~pol].max() <= 5e-2
    lo = bits_to_rows(r.active_lower.cpu().numpy(), het.m_ref); up = bits_to_rows(r.active_upper.cpu().numpy(), het.m_ref)
    for b in np.nonzero(pol)[0]:
        assert np.array_equal(lo[b], oact[b][0]) and np.array_equal(up[b], oact[b][1]), b
    print("4096 distinct controllers: %d polished by the oracle, worst relative error of u* %.2e, of the cost %.2e" % (pol.sum(), err[pol].max(), cerr[pol].max()))
