"""The NLMPC solve kernels stepped through on the host by the lock-step interpreter of tests/emu (TEST INFRASTRUCTURE: fibres per thread,
rendezvous for DPP / readlane / shuffles / barriers -- see tests/emu/hip/hip_runtime.h): the device sources of include/mpcx/ are compiled
unchanged with g++ and must reach the oracle's optimum, in both forms of the kernel and in both orders in which the interpreter may run the
threads of a workgroup between two rendezvous (an exchange through LDS that lacks its barrier gives different results in the two orders,
a wave-level operation inside divergent control flow a reported deadlock).  No GPU, nothing of libmpcx.so."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import nlmpc_numpy as ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu")


@pytest.fixture(scope="module")
def runner(tmp_path_factory):
    if not shutil.which("g++"):
        pytest.skip("g++ not installed")
    exe = str(tmp_path_factory.mktemp("emu") / "run_nlmpc")
    subprocess.run(["g++", "-O1", "-std=c++20", "-DHIPEMU_WITH_WG", "-I" + EMU, "-I" + os.path.join(ROOT, "include"), "-fpermissive", "-w", "-o", exe,
                    os.path.join(EMU, "run_nlmpc.cpp"), os.path.join(EMU, "hipemu_switch.S")], check=True)

    def run(args, inst, env=None):
        e = dict(os.environ); e.update(env or {})
        inp = "\n".join(" ".join(repr(float(x)) for x in row) for row in inst) + "\n"
        r = subprocess.run([exe] + [str(a) for a in args], input=inp, capture_output=True, text=True, env=e, timeout=600)
        assert r.returncode == 0, r.stderr[:2000]
        return [json.loads(l) for l in r.stdout.splitlines()]
    return run


@pytest.mark.parametrize("form,env", [("wg", {}), ("wg", {"HIPEMU_ORDER": "reverse"}), ("wg", {"HIPEMU_WAVES": "4"}), ("wave", {}), ("wave", {"HIPEMU_ORDER": "reverse"})])
def test_vanderpol_reaches_the_oracle_optimum_through_the_interpreter(runner, form, env):
    rng = np.random.default_rng(11)
    X0 = rng.uniform(-1.0, 1.0, size=(4, 2)); X0[0] = [0.0, 1.0]          # examples/vanderpol_ex.cpp:67
    r = runner(["vanderpol", 10, 5, 0.1, 1, 200, form], np.hstack([X0, np.zeros((4, 1))]), env)
    m = ref.vanderpol(ph=10, ch=5, Ts=0.1)
    for b, y in enumerate(r):
        o = m.solve(X0[b], np.zeros(1), max_iter=1000)
        assert o["success"] and y["status"] == 0 and y["solver_status"] == 4
        np.testing.assert_allclose(y["cmd"], o["cmd"], rtol=1e-5, atol=1e-5)
        assert abs(y["cost"] - o["cost"]) <= 1e-8 * max(1.0, abs(o["cost"]))


@pytest.mark.parametrize("env", [{}, {"HIPEMU_ORDER": "reverse"}])
def test_workgroup_form_with_bounds_equalities_and_dense_rows(runner, env):
    """rows through the sensitivities (state bounds, the UGV's obstacle rows, a terminal equality), sparse rows (input bounds), infeasible
    starts: the workgroup form against the one-wavefront form, instance by instance"""
    rng = np.random.default_rng(21)
    X0 = rng.uniform(-0.7, 0.7, size=(6, 2)); X0[0] = [0.0, 1.0]
    cases = [(["vanderpol", 10, 5, 0.1, 1, 200], np.hstack([X0, np.zeros((6, 1))]), ["lbu=-0.3", "ubu=0.3", "lbx0=-0.8", "ubx0=0.8"]),
             (["vanderpol_terminal", 10, 5, 0.1, 1, 300], np.hstack([0.15 * X0[:3], np.zeros((3, 1))]), []),
             (["ugv", 12, 4, 0.1, 0, 150], np.hstack([np.c_[0.4 * X0[:3], np.zeros((3, 2))], np.zeros((3, 2))]), [])]
    for args, inst, extra in cases:
        a = runner(args + ["wave"] + extra, inst)
        # (the same route: both from the identity -- the workgroup form's Gauss-Newton start is the next test's subject)
        b = runner(args + ["wg"] + extra, inst, dict(env, MPCX_NLMPC_CURV0="0"))
        for x, y in zip(a, b):
            assert x["status"] == y["status"] and x["solver_status"] == y["solver_status"], (args[0], x["b"])
            if x["status"] == 0:
                np.testing.assert_allclose(y["cmd"], x["cmd"], rtol=1e-5, atol=1e-5)
                assert abs(x["iterations"] - y["iterations"]) <= 3
        c = runner(args + ["wg"] + extra, inst, env)            # ... and as the library runs it: the same statuses and optimum
        for x, y in zip(a, c):
            assert x["status"] == y["status"], (args[0], x["b"])
            if x["status"] == 0:
                np.testing.assert_allclose(y["cmd"], x["cmd"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("env", [{}, {"HIPEMU_ORDER": "reverse"}])
def test_gauss_newton_start_of_the_curvature_estimate(runner, env):
    """WgSqp::init_curvature (the condensed Hessian of the cost, sixteen-row tiles in the MFMA accumulators, inverted in place): the same optimum as
    from the identity in fewer iterations, the same bits in both thread orders (a missing barrier in the new phase would show), at the start of
    the solve (oscillators, Van der Pol) and installed after ten iterations (the UGV: Ugv::CURV0_AFTER)"""
    x6 = np.zeros((2, 18)); x6[:, 0] = 1.0; x6[1, 1:12] = np.linspace(-0.1, 0.1, 11)
    cases = [(["vanderpol", 10, 5, 0.1, 1, 200], np.array([[0.0, 1.0, 0.0], [0.5, -0.4, 0.0]])),
             (["ugv", 30, 30, 0.1, 0, 150], np.array([[0.1, -0.2, 0, 0, 0, 0], [-0.3, 0.4, 0, 0, 0, 0]], float)),
             (["osc6", 20, 10, 0.1, 1, 200], x6)]
    for args, inst in cases:
        a = runner(args + ["wg"], inst, dict(env, MPCX_NLMPC_CURV0="0"))
        b = runner(args + ["wg"], inst, env)
        for x, y in zip(a, b):
            assert x["status"] == 0 and y["status"] == 0
            np.testing.assert_allclose(y["cmd"], x["cmd"], rtol=1e-5, atol=1e-5)
            assert abs(y["cost"] - x["cost"]) <= 1e-8 * abs(x["cost"])
            assert y["iterations"] < x["iterations"] or x["iterations"] <= 6, (args[0], x["iterations"], y["iterations"])
        if env:
            f = runner(args + ["wg"], inst, {})
            for x, y in zip(f, b):
                assert x["cmd"] == y["cmd"] and x["cost"] == y["cost"] and x["iterations"] == y["iterations"]


def test_config3_golden_instances_through_the_interpreter(runner):
    """the first golden instances of BASELINE config 3 (tests/golden/nlmpc_oracle_solutions.json) in the workgroup form, four wavefronts"""
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "nlmpc_oracle_solutions.json")))["ugv_ph30_ch30"]
    cases = gold["cases"][1:5]
    r = runner(["ugv", 30, 30, 0.1, 0, 150, "wg"], np.array([k["x0"] + k["u0"] for k in cases]))
    for k, y in zip(cases, r):
        assert y["status"] == 0
        np.testing.assert_allclose(y["cmd"], k["cmd"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("env", [{"HIPEMU_BLOCKS": "0"}, {"HIPEMU_BLOCKS": "0", "HIPEMU_ORDER": "reverse"}])
def test_config3_with_blocks_and_reduced_rows_in_the_workspace(runner, env):
    """the variant of the workgroup form that keeps the folded blocks and the reduced rows in the per-instance workspace (three workgroups per CU
    at config 3; its reduced rows are stored, not added, by the condensing sweep): golden instances, both thread orders"""
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "nlmpc_oracle_solutions.json")))["ugv_ph30_ch30"]
    cases = gold["cases"][1:3]
    inst = np.array([k["x0"] + k["u0"] for k in cases])
    r = runner(["ugv", 30, 30, 0.1, 0, 150, "wg"], inst, env)
    for k, y in zip(cases, r):
        assert y["status"] == 0
        np.testing.assert_allclose(y["cmd"], k["cmd"], rtol=1e-5, atol=1e-5)
    # ... and with a bound on a state from the fourth state row on (rows of the bounds through the sweep as well)
    a = runner(["ugv", 12, 4, 0.1, 0, 150, "wg", "lbx0=-0.6", "ubx0=0.9", "xs=3", "lbu=-3", "ubu=3"], inst, env)
    b = runner(["ugv", 12, 4, 0.1, 0, 150, "wave", "lbx0=-0.6", "ubx0=0.9", "xs=3", "lbu=-3", "ubu=3"], inst)
    for x, y in zip(a, b):
        assert x["status"] == y["status"]
        if x["status"] == 0:
            np.testing.assert_allclose(x["cmd"], y["cmd"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("env", [{}, {"HIPEMU_ORDER": "reverse"}])
def test_lds_blocks_next_to_a_two_level_factor_equal_the_workspace_form(runner, env):
    """nlmpc_sqp<Model, true, true> (dynamics blocks and defects in LDS next to a two-level working-set factor): the combination that returned
    wrong results in an experiment of round 3 and that the plan never selects.  The source is race-free: both thread
    orders give the workspace form's results (on the GPU the probe build is bit-identical to it, profiles/r04_probe_blk_two_level.txt)"""
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "nlmpc_oracle_solutions.json")))["ugv_ph30_ch30"]
    cases = gold["cases"][1:3]
    inst = np.array([k["x0"] + k["u0"] for k in cases])
    a = runner(["ugv", 30, 30, 0.1, 0, 150, "wave"], inst)
    b = runner(["ugv", 30, 30, 0.1, 0, 150, "wave-blk2"], inst, env)
    for x, y in zip(a, b):
        assert x["status"] == y["status"] == 0 and x["iterations"] == y["iterations"]
        np.testing.assert_allclose(y["cmd"], x["cmd"], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("env", [{}, {"HIPEMU_ORDER": "reverse"}])
def test_inverse_of_the_schur_complement_equals_the_factor_form(runner, env):
    """WgPlan::minv (round 5): the working set's Schur complement kept as its inverse -- product, bordering and rank-one correction over the
    workgroup, one row shed per warm-start round -- against the Cholesky form of the same kernel on the six-oscillator network (working
    sets of 20 to 50 input bounds), in both thread orders: same statuses, the same optimum to the solver's own tolerance"""
    rng = np.random.default_rng(5)
    X0 = rng.uniform(-0.1, 0.1, size=(2, 12)); X0[:, 0] += 1.0
    inst = np.hstack([X0, np.zeros((2, 6))])
    a = runner(["osc6", 20, 10, 0.1, 1, 200, "wg"], inst, {"MPCX_NLMPC_MINV": "0"})
    e = dict(env); e["MPCX_NLMPC_MINV"] = "1"
    b = runner(["osc6", 20, 10, 0.1, 1, 200, "wg"], inst, e)
    for x, y in zip(a, b):
        assert x["status"] == y["status"] == 0
        np.testing.assert_allclose(y["cmd"], x["cmd"], rtol=1e-5, atol=1e-6)
        assert abs(x["cost"] - y["cost"]) <= 1e-9 * max(1.0, abs(x["cost"]))
        assert y["dual_steps"] <= x["dual_steps"]            # (shedding one row at a time keeps more of the kept set)


def test_eight_wavefronts_per_instance_and_the_mfma_schur_complement(runner):
    """the eight-wavefront variant (systems alone on a CU) on a small oscillator problem, and config 3's first golden instance with the
    dense rows' Schur complement on v_mfma_f64_16x16x4 (the interpreter models the instruction lane for lane)"""
    rng = np.random.default_rng(7)
    X0 = rng.uniform(-0.1, 0.1, size=(1, 12)); X0[:, 0] += 1.0
    inst = np.hstack([X0, np.zeros((1, 6))])
    a = runner(["osc6", 12, 6, 0.1, 1, 200, "wg"], inst, {"HIPEMU_WAVES": "4"})
    b = runner(["osc6", 12, 6, 0.1, 1, 200, "wg"], inst, {"HIPEMU_WAVES": "8"})
    assert a[0]["status"] == b[0]["status"] == 0
    np.testing.assert_allclose(b[0]["cmd"], a[0]["cmd"], rtol=1e-5, atol=1e-6)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "nlmpc_oracle_solutions.json")))["ugv_ph30_ch30"]
    k = gold["cases"][1]
    y = runner(["ugv", 30, 30, 0.1, 0, 150, "wg"], np.array([k["x0"] + k["u0"]]), {"HIPEMU_BLOCKS": "1"})[0]
    assert y["status"] == 0 and y["max_nw"] >= 2              # a kept set with dense rows went through ws_schur_mfma
    np.testing.assert_allclose(y["cmd"], k["cmd"], rtol=1e-5, atol=1e-5)


def test_carried_inverse_stays_the_inverse(tmp_path_factory):
    """WgPlan::carry_m (round 5): where every row of the sub-problem has the same entries at every iterate (bounds on inputs, user rows affine in
    the inputs -- Mdl::XFREE_ROWS_AFFINE, their entries differenced once), the inverse of the working set's Schur complement is carried from one
    sub-problem to the next through the rank-two change of B^-1 (Woodbury) instead of being formed by n sweeps.  A build with
    -DHIPEMU_CHECK_CARRY forms S afresh at every carried warm start and reports |S M - I|; the solve must be the one without carrying."""
    if not shutil.which("g++"):
        pytest.skip("g++ not installed")
    exe = str(tmp_path_factory.mktemp("emuchk") / "run_chk")
    subprocess.run(["g++", "-O1", "-std=c++20", "-DHIPEMU_WITH_WG", "-DHIPEMU_CHECK_CARRY", "-I" + EMU, "-I" + os.path.join(ROOT, "include"), "-fpermissive", "-w",
                    "-o", exe, os.path.join(EMU, "run_nlmpc.cpp"), os.path.join(EMU, "hipemu_switch.S")], check=True)
    rng = np.random.default_rng(5)
    X0 = rng.uniform(-0.1, 0.1, size=(1, 12)); X0[:, 0] += 1.0
    inp = " ".join(repr(float(x)) for x in np.hstack([X0, np.zeros((1, 6))])[0]) + "\n"
    out = {}
    for carry, order in (("0", "forward"), ("1", "forward"), ("1", "reverse")):
        # (from the identity: thirty sub-problems to carry the inverse through; from the Gauss-Newton start the solve takes nine)
        e = dict(os.environ); e.update({"MPCX_NLMPC_MINV": "1", "MPCX_NLMPC_CARRY": carry, "HIPEMU_ORDER": order, "MPCX_NLMPC_CURV0": "0"})
        r = subprocess.run([exe, "osc6", "20", "10", "0.1", "1", "200", "wg"], input=inp, capture_output=True, text=True, env=e, timeout=900)
        assert r.returncode == 0, r.stderr[:2000]
        errs = [float(l.split("=")[-1]) for l in r.stderr.splitlines() if l.startswith("carry check")]
        out[(carry, order)] = (json.loads(r.stdout.splitlines()[0]), errs)
    plain, _ = out[("0", "forward")]
    assert plain["carried"] == 0
    for key in (("1", "forward"), ("1", "reverse")):
        y, errs = out[key]
        assert y["carried"] >= 20 and len(errs) == y["carried"] and max(errs) <= 1e-11, (y["carried"], max(errs))
        assert y["status"] == plain["status"] == 0 and y["iterations"] == plain["iterations"] and y["dual_steps"] == plain["dual_steps"]
        assert abs(y["cost"] - plain["cost"]) <= 1e-12 * max(1.0, abs(plain["cost"]))
        np.testing.assert_allclose(y["cmd"], plain["cmd"], rtol=1e-6, atol=1e-7)
    assert out[("1", "forward")][0]["cmd"] == out[("1", "reverse")][0]["cmd"]           # (no exchange through LDS without its barrier)


@pytest.mark.parametrize("env", [{}, {"HIPEMU_ORDER": "reverse"}, {"MPCX_NLMPC_MINV": "1"}])
def test_rows_with_two_entries_and_several_rows_on_one_input(runner, env):
    """mpcx::models::VanDerPolRate (|u_i - u_{i-1}| <= 0.1 next to u_i <= 0.5): short-list rows with two entries, several of them on one input -- N_W' r as
    the fixed-order gather (WgSqp::sparse_gather) in the factor form and, forced, in the inverse form; both thread orders; against the oracle"""
    rng = np.random.default_rng(3)
    X0 = rng.uniform(-1.0, 1.0, size=(4, 2)); X0[0] = [0.0, 1.0]
    r = runner(["vanderpol_rate", 10, 10, 0.1, 1, 200, "wg"], np.hstack([X0, np.zeros((4, 1))]), env)
    m = ref.vanderpol_rate(ph=10, ch=10, Ts=0.1, rate=0.1)
    for b, y in enumerate(r):
        o = m.solve(X0[b], np.zeros(1), max_iter=1000)
        assert o["success"] and y["status"] == 0
        np.testing.assert_allclose(y["cmd"], o["cmd"], rtol=1e-5, atol=1e-5)
        assert abs(y["cost"] - o["cost"]) <= 1e-8 * max(1.0, abs(o["cost"]))
