"""The C++20 front-end (include/mpc/LMPC.hpp -> include/mpcx/LMPC.hpp -> C ABI): the reference's LMPC
test scenarios, compiled with g++ against this repository's headers."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "lmpc_frontend_test.cpp")
OUT = os.path.join(ROOT, "tests", "cpp", "build", "lmpc_frontend_test")


def _build(src=SRC, out=OUT):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    lib = os.path.join(ROOT, "libmpc_amd")
    cmd = ["g++", "-std=c++20", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), src, "-o", out,
           "-L" + lib, "-lmpcx", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return out


def test_cpp_frontend_api_without_gpu():
    exe = _build()
    env = dict(os.environ, MPCX_DEVICE="-1")
    out = subprocess.run([exe, "api"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all C++ front-end checks passed" in out.stdout


@pytest.mark.gpu
def test_cpp_frontend_quadrotor_known_answer():
    """reference test/LMPC/test_common.cpp:89-237, static and dynamic sizes, through the C++ header"""
    exe = _build()
    env = {k: v for k, v in os.environ.items() if k != "MPCX_DEVICE"}
    out = subprocess.run([exe, "solve"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all C++ front-end checks passed" in out.stdout


NL_SRC = os.path.join(ROOT, "tests", "cpp", "nlmpc_frontend_test.cpp")
NL_OUT = os.path.join(ROOT, "tests", "cpp", "build", "nlmpc_frontend_test")


def test_cpp_nlmpc_frontend_api_without_gpu():
    exe = _build(NL_SRC, NL_OUT)
    out = subprocess.run([exe, "api"], env=dict(os.environ, MPCX_DEVICE="-1"), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all C++ NLMPC front-end checks passed" in out.stdout


@pytest.mark.gpu
def test_cpp_nlmpc_frontend_vanderpol_closed_loop():
    """reference examples/vanderpol_ex.cpp through include/mpc/NLMPC.hpp: closed loop to the origin, first move = the oracle's"""
    exe = _build(NL_SRC, NL_OUT)
    env = {k: v for k, v in os.environ.items() if k != "MPCX_DEVICE"}
    out = subprocess.run([exe, "solve"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all C++ NLMPC front-end checks passed" in out.stdout


BK_SRC = os.path.join(ROOT, "tests", "cpp", "ioptimizer_backend_test.cpp")
BK_OUT = os.path.join(ROOT, "tests", "cpp", "build", "ioptimizer_backend_test")


def _build_backend(src=BK_SRC, out=BK_OUT):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    lib = os.path.join(ROOT, "libmpc_amd")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                           src, "-o", out, "-L" + lib, "-lmpcx", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    return out


def test_ioptimizer_shaped_backend_api_without_gpu():
    """INTEGRATION.md section 2's binding (a backend of the shape of IOptimizer.hpp:24-58 over the C ABI), compiled"""
    exe = _build_backend()
    out = subprocess.run([exe, "api"], env=dict(os.environ, MPCX_DEVICE="-1"), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all IOptimizer backend checks passed" in out.stdout


@pytest.mark.gpu
def test_ioptimizer_shaped_backend_solves_like_the_front_end():
    exe = _build_backend()
    env = {k: v for k, v in os.environ.items() if k != "MPCX_DEVICE"}
    out = subprocess.run([exe, "solve"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all IOptimizer backend checks passed" in out.stdout


NLBK_SRC = os.path.join(ROOT, "tests", "cpp", "nloptimizer_backend_test.cpp")
NLBK_OUT = os.path.join(ROOT, "tests", "cpp", "build", "nloptimizer_backend_test")


def test_nloptimizer_shaped_backend_api_without_gpu():
    """INTEGRATION.md section 6's binding (a backend of the shape of NLOptimizer.hpp:30-404 over the C ABI), compiled; its hook sources
    cross-compile for gfx950 without a device"""
    if not os.path.exists("/opt/rocm/lib/libhiprtc.so") and not any(f.startswith("libhiprtc.so") for f in os.listdir("/opt/rocm/lib")):
        pytest.skip("hipRTC not installed")
    exe = _build_backend(NLBK_SRC, NLBK_OUT)
    out = subprocess.run([exe, "api"], env=dict(os.environ), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all NLOptimizer backend checks passed" in out.stdout


@pytest.mark.gpu
def test_nloptimizer_shaped_backend_solves_like_the_front_end():
    exe = _build_backend(NLBK_SRC, NLBK_OUT)
    env = {k: v for k, v in os.environ.items() if k != "MPCX_DEVICE"}
    out = subprocess.run([exe, "solve"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all NLOptimizer backend checks passed" in out.stdout
