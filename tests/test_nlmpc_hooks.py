"""User hooks of the NLMPC path (NLMPC::setStateSpaceFunction & co., reference NLMPC.hpp:139-281) as device code:
compiled at run time from the lambda bodies (mpcx_nlmpc_create_from_source), compiled by hipcc in the caller's
translation unit (include/mpcx/nlmpc_hooks.hpp through mpc::NLMPC<>'s setters).  CPU part: both compile for gfx950
without a GPU; GPU part: they reproduce the built-in models."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VDP = dict(state_fn="dx(0) = ((1.0 - (x(1) * x(1))) * x(0)) - x(1) + u(0); dx(1) = x(0);",         # examples/vanderpol_ex.cpp:38-39
           objective_fn="return x.array().square().sum() + u.array().square().sum();",                # :54
           ineq_fn="for (int i = 0; i < ineq_c; i++) { in_con(i) = u(i, 0) - 0.5; }")                 # :62-64

# examples/ugv_ex.cpp:32-124 with the constants its closures capture spelled in the preamble
UGV_PRE = """
__device__ inline mpc::cvec<2> v_pref() { mpc::cvec<2> v; v(0) = 0.7071067811865476; v(1) = 0.7071067811865476; return v; }
struct Obstacle { double px, py, radius; };
__device__ const Obstacle obs[2] = {{2.0, 1.0, 0.3}, {1.0, 1.0, 0.3}};
constexpr double Ts = 0.1;
"""
UGV = dict(preamble=UGV_PRE,
           state_fn="""dx(0) = x(0) + Ts * x(2) + 0.5 * Ts * Ts * u(0); dx(1) = x(1) + Ts * x(3) + 0.5 * Ts * Ts * u(1);
                       dx(2) = x(2) + Ts * u(0); dx(3) = x(3) + Ts * u(1);""",
           output_fn="y = x;",
           objective_fn="""double cost = 0;
                           for (int i = 0; i < pred_hor + 1; i++) {
                               cost += 1e3 * (x.row(i).segment(2, 2).transpose() - v_pref()).squaredNorm();
                               cost += 1e-2 * u.row(i).squaredNorm();
                           }
                           cost += 1e-5 * e * e;
                           return cost;""",
           ineq_fn="""int index = 0;
                      for (int i = 0; i < pred_hor + 1; i++)
                          for (int j = 0; j < 2; j++) {
                              const double rx = x(i, 0) - obs[j].px, ry = x(i, 1) - obs[j].py;
                              in_con(index++) = obs[j].radius - sqrt(rx * rx + ry * ry);
                          }""")


def _source(dims, hooks):
    from libmpc_amd import _capi
    enc = lambda t: None if t is None else t.encode()
    return _capi.NlmpcSource(*dims, enc(hooks.get("preamble")), enc(hooks["state_fn"]), enc(hooks["objective_fn"]),
                             enc(hooks.get("ineq_fn")), enc(hooks.get("eq_fn")), enc(hooks.get("output_fn")))


def test_hook_sources_compile_for_gfx950_without_a_gpu():
    """hipRTC needs no device to compile: the generated translation unit (engine + hooks) builds, and a broken hook
    comes back as MPCX_E_INVALID with the compiler's diagnostics"""
    from libmpc_amd import _capi
    lib = _capi.lib()
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/lib/libhiprtc.so.7"):
        pytest.skip("hipRTC not available")
    n = lib.mpcx_nlmpc_debug_compile_source(C.byref(_source((2, 1, 2, 10, 5, 11, 0), VDP)))
    assert n > 10000, lib.mpcx_last_error()
    n = lib.mpcx_nlmpc_debug_compile_source(C.byref(_source((4, 2, 4, 12, 4, 26, 0), UGV)))
    assert n > 10000, lib.mpcx_last_error()
    bad = dict(VDP, objective_fn="return x.array().square().sum() + no_such_symbol;")
    rc = lib.mpcx_nlmpc_debug_compile_source(C.byref(_source((2, 1, 2, 10, 5, 11, 0), bad)))
    assert rc == _capi.E_INVALID and b"no_such_symbol" in lib.mpcx_last_error()
    # argument checks
    assert lib.mpcx_nlmpc_debug_compile_source(C.byref(_source((2, 1, 2, 10, 5, 0, 0), VDP))) == _capi.E_INVALID     # ineq_fn without rows
    assert lib.mpcx_nlmpc_debug_compile_source(C.byref(_source((2, 1, 2, 5, 10, 11, 0), VDP))) == _capi.E_INVALID     # ch > ph


HOOKS_SRC = os.path.join(ROOT, "tests", "cpp", "nlmpc_hooks_test.cpp")
HOOKS_OUT = os.path.join(ROOT, "tests", "cpp", "build", "nlmpc_hooks_test")


def _build_hooks_test():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    os.makedirs(os.path.dirname(HOOKS_OUT), exist_ok=True)
    lib = os.path.join(ROOT, "libmpc_amd")
    if not os.path.exists(HOOKS_OUT) or os.path.getmtime(HOOKS_OUT) < max(os.path.getmtime(HOOKS_SRC), os.path.getmtime(os.path.join(lib, "libmpcx.so"))):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-std=c++20", "-O3", "-I" + os.path.join(ROOT, "include"), HOOKS_SRC,
                               "-o", HOOKS_OUT, "-L" + lib, "-lmpcx", "-Wl,-rpath," + lib])
    return HOOKS_OUT


def test_reference_lambdas_compile_against_the_front_end():
    """examples/vanderpol_ex.cpp's setter calls, lambdas with the reference's parameter lists and bodies, build with hipcc"""
    assert os.path.exists(_build_hooks_test())


@pytest.mark.gpu
def test_reference_lambdas_through_the_setters_match_the_builtin_model():
    exe = _build_hooks_test()
    env = {k: v for k, v in os.environ.items() if k != "MPCX_DEVICE"}
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=600)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all C++ NLMPC hook checks passed" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["vanderpol", "ugv"])
def test_hooks_from_sources_match_the_builtin_model(name):
    """the same transcription and the same optimum whether the system is built in or comes as source text"""
    import torch
    from libmpc_amd.nlmpc import NLMPC, NLParameters, UGV as UGV_ID, VANDERPOL
    if name == "vanderpol":
        zoo = NLMPC(VANDERPOL, 10, 5, 0.1)
        usr = NLMPC.from_sources(2, 1, 2, 10, 5, 11, 0, 0.1, **VDP)
        hard, iters = 1, 200
    else:
        zoo = NLMPC(UGV_ID, 12, 4, 0.1)
        usr = NLMPC.from_sources(4, 2, 4, 12, 4, 26, 0, 0.0, **UGV)
        hard, iters = 0, 150
    assert (usr.nz, usr.nineq, usr.neq_user) == (zoo.nz, zoo.nineq, zoo.neq_user)
    rng = np.random.default_rng(5)
    B = 7
    Z = torch.from_numpy(rng.normal(size=(B, zoo.nz))); X0 = torch.from_numpy(rng.normal(size=(B, zoo.nx)))
    a = zoo.evaluate(Z, X0); b = usr.evaluate(Z, X0)
    torch.cuda.synchronize()
    for k in ("cost", "grad", "ceq", "jeq", "cineq", "jineq"):
        np.testing.assert_allclose(b[k].cpu().numpy(), a[k].cpu().numpy(), rtol=1e-9, atol=1e-6 if k.startswith("j") or k == "grad" else 1e-12, err_msg=k)
    for c in (zoo, usr):
        c.setOptimizerParameters(NLParameters(maximum_iteration=iters, hard_constraints=hard))
    x0 = np.zeros((B, zoo.nx)); x0[:, :2] = rng.uniform(-0.5, 0.5, size=(B, 2))
    if name == "vanderpol":
        x0[0] = [0.0, 1.0]
    u0 = np.zeros((B, zoo.nu))
    ra = zoo.optimizeBatch(torch.from_numpy(x0), torch.from_numpy(u0), sequences=True)
    rb = usr.optimizeBatch(torch.from_numpy(x0), torch.from_numpy(u0), sequences=True)
    torch.cuda.synchronize()
    assert (ra["status"].cpu().numpy() != 3).all() and np.array_equal(ra["status"].cpu().numpy(), rb["status"].cpu().numpy())
    np.testing.assert_allclose(rb["cmd"].cpu().numpy(), ra["cmd"].cpu().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(rb["cost"].cpu().numpy(), ra["cost"].cpu().numpy(), rtol=1e-9)
    if name == "ugv":                                       # y = x through the output function
        np.testing.assert_allclose(rb["seq_output"].cpu().numpy(), rb["seq_state"].cpu().numpy(), rtol=0, atol=0)


@pytest.mark.gpu
def test_hooks_with_user_equalities_and_an_output_function_match_the_builtin_model():
    """setEqConFunction and setOutputFunction as sources: Van der Pol with the terminal equality x(ph) = 0, the cost written
    on the outputs y = x (Model::getOutput is re-run inside every perturbation, Model.hpp:72-96)"""
    import torch
    from libmpc_amd.nlmpc import NLMPC, NLParameters, VANDERPOL_TERMINAL
    zoo = NLMPC(VANDERPOL_TERMINAL, 10, 5, 0.1)
    usr = NLMPC.from_sources(2, 1, 2, 10, 5, 11, 2, 0.1, state_fn=VDP["state_fn"], ineq_fn=VDP["ineq_fn"],
                             output_fn="y(0) = x(0); y(1) = x(1);",
                             objective_fn="return y.array().square().sum() + u.array().square().sum();",
                             eq_fn="eq_con(0) = x(pred_hor, 0); eq_con(1) = x(pred_hor, 1);")
    assert usr.neq_user == 2
    rng = np.random.default_rng(8)
    B = 6
    Z = torch.from_numpy(rng.normal(size=(B, zoo.nz))); X0 = torch.from_numpy(rng.normal(size=(B, 2)))
    a = zoo.evaluate(Z, X0); b = usr.evaluate(Z, X0)
    torch.cuda.synchronize()
    for k in ("cost", "ceq", "jeq", "cineq", "jineq", "grad"):
        # the forward-difference gradient amplifies one ulp of the cost by 1 / 1.5e-8
        atol = 1e-4 if k == "grad" else (1e-6 if k.startswith("j") else 1e-12)
        np.testing.assert_allclose(b[k].cpu().numpy(), a[k].cpu().numpy(), rtol=1e-9, atol=atol, err_msg=k)
    for c in (zoo, usr):
        c.setOptimizerParameters(NLParameters(maximum_iteration=300))
    x0 = rng.uniform(-0.12, 0.12, size=(B, 2)); x0[0] = [0.1, 0.1]
    u0 = np.zeros((B, 1))
    ra = zoo.optimizeBatch(torch.from_numpy(x0), torch.from_numpy(u0), sequences=True)
    rb = usr.optimizeBatch(torch.from_numpy(x0), torch.from_numpy(u0), sequences=True)
    torch.cuda.synchronize()
    sa, sb = ra["solver_status"].cpu().numpy(), rb["solver_status"].cpu().numpy()
    print("solver status", sa, sb, "iterations", ra["iterations"].cpu().numpy(), rb["iterations"].cpu().numpy())
    ok = sa == 4
    assert np.array_equal(sa, sb) and ok[0] and ok.sum() >= B - 2, (sa, sb)      # a random start may be unable to reach the origin: both say so
    np.testing.assert_allclose(rb["cmd"].cpu().numpy()[ok], ra["cmd"].cpu().numpy()[ok], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(rb["seq_output"].cpu().numpy(), rb["seq_state"].cpu().numpy(), rtol=0, atol=0)
    assert np.abs(rb["seq_state"].cpu().numpy()[ok][:, 10]).max() <= 1e-9


@pytest.mark.gpu
def test_python_callables_are_refused_with_a_pointer_to_sources():
    from libmpc_amd.nlmpc import NLMPC, VANDERPOL
    c = NLMPC(VANDERPOL, 10, 5, 0.1)
    with pytest.raises(RuntimeError, match="from_sources"):
        c.setObjectiveFunction(lambda X, Y, U, e: 0.0)
