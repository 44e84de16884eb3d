"""CPU tests of the product's host side (no GPU): C-ABI surface, setter semantics of the
reference front-end, the condensing, and the refusal to solve without a device."""
import os
import re

import numpy as np
import pytest

from helpers import INF, OracleFrontEnd, configure_quadrotor, configure_random, random_lmpc_spec
from oracle import lmpc_numpy as LN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    from libmpc_amd import _capi
    hdr = open(os.path.join(ROOT, "include", "mpcx.h")).read()
    declared = set(re.findall(r"\b(mpcx_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = _capi.lib()
    for name in sorted(declared):
        getattr(lib, name)
    assert declared <= set(_capi.EXPORTS) | {"mpcx_version", "mpcx_last_error"}
    assert lib.mpcx_version().startswith(b"mpcx")


def test_cabi_declares_every_exported_symbol():
    """the other direction: whatever the library exports under the mpcx_ prefix is declared in include/mpcx.h -- the measured and
    tested ABI is the documented one"""
    import shutil
    import subprocess
    from libmpc_amd import _capi
    nm = shutil.which("nm")
    if not nm:
        pytest.skip("binutils not installed")
    out = subprocess.run([nm, "-D", "--defined-only", _capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("mpcx_")}
    hdr = open(os.path.join(ROOT, "include", "mpcx.h")).read()
    declared = set(re.findall(r"\b(mpcx_[a-z0-9_]+)\s*\(", hdr))
    assert exported - declared == set(), sorted(exported - declared)


def test_unsupported_calls_raise_like_the_reference():
    """LMPC.hpp:68-100: discrete-time only, no scaling"""
    from libmpc_amd import LMPC
    c = LMPC(2, 1, 0, 2, 5, 5, device=-1)
    with pytest.raises(RuntimeError):
        c.setDiscretizationSamplingTime(0.1)
    with pytest.raises(RuntimeError):
        c.setInputScale([1.0])
    with pytest.raises(RuntimeError):
        c.setStateScale([1.0, 1.0])


def test_slice_validation_and_return_values():
    """test/LMPC/test_lmpc.cpp: setters return true; invalid slices return false (IMPC.hpp:244-283)"""
    from libmpc_amd import LMPC
    nx, nu, ndu, ny, ph, ch = 3, 2, 1, 2, 6, 4
    c = LMPC(nx, nu, ndu, ny, ph, ch, device=-1)
    assert c.setStateSpaceModel(np.eye(nx), np.ones((nx, nu)), np.ones((ny, nx)))
    assert c.setDisturbances(np.zeros((nx, ndu)), np.zeros((ny, ndu)))
    assert c.setObjectiveWeights(np.ones((ny, ph)), np.ones((nu, ph)), np.ones((nu, ph)))
    assert c.setObjectiveWeights(np.ones(ny), np.ones(nu), np.ones(nu), (0, ph))
    assert c.setObjectiveWeights(np.ones(ny), np.ones(nu), np.ones(nu), (-1, -1))
    assert not c.setObjectiveWeights(np.ones(ny), np.ones(nu), np.ones(nu), (3, 3))
    assert not c.setObjectiveWeights(np.ones(ny), np.ones(nu), np.ones(nu), (0, ph + 1))
    assert c.setStateBounds(-np.ones((nx, ph)), np.ones((nx, ph)))
    assert c.setStateBounds(-np.ones(nx), np.ones(nx), (0, 1))
    assert not c.setStateBounds(-np.ones(nx), np.ones(nx), (2, 1))
    assert c.setInputBounds(-np.ones((nu, ch)), np.ones((nu, ch)))
    assert c.setInputBounds(-np.ones(nu), np.ones(nu), (0, ch))
    assert not c.setInputBounds(-np.ones(nu), np.ones(nu), (0, ch + 1))      # control-horizon slice
    assert c.setOutputBounds(-np.ones(ny), np.ones(ny), (0, ph))
    assert c.setScalarConstraint(-INF, INF, np.ones(nx), np.ones(nu), (-1, -1))
    assert c.setScalarConstraint(0, -INF, INF, np.ones(nx), np.ones(nu))
    assert not c.setScalarConstraint(ph, -1.0, 1.0, np.ones(nx), np.ones(nu))
    assert c.setReferences(np.zeros((ny, ph)), np.zeros((nu, ph)), np.zeros((nu, ph)))
    assert c.setReferences(np.zeros(ny), np.zeros(nu), np.zeros(nu), (0, ph))
    assert c.setExogenousInputs(np.zeros((ndu, ph)))
    # the reference validates this slice against the CONTROL horizon (LMPC.hpp:571: isControlHorizonSliceValid)
    assert c.setExogenousInputs(np.zeros(ndu), (0, ch))
    assert c.setExogenousInputs(np.zeros(ndu), (-1, -1))
    assert c.setExogenousInputs(np.zeros(ndu), (0, ph)) == (ph <= ch)
    with pytest.raises(ValueError):
        c.setStateSpaceModel(np.eye(nx + 1), np.ones((nx, nu)), np.ones((ny, nx)))


def test_no_cpu_solve_path():
    from libmpc_amd import LMPC, MpcxError
    from libmpc_amd.workloads import quadrotor_lmpc
    c = quadrotor_lmpc(10, device=-1)
    with pytest.raises(MpcxError):
        c.optimize(np.zeros(12), np.zeros(4))
    bad = LMPC(2, 1, 0, 2, 5, 5, device=-1)
    with pytest.raises(MpcxError):       # no model yet
        bad.setup()
    with pytest.raises(MpcxError):
        LMPC(80, 1, 0, 2, 5, 5, device=-1)      # one lane per state component: nx <= 64


@pytest.mark.parametrize("ph", [10, 20, 50])
def test_reference_qp_sizes(ph):
    from libmpc_amd.workloads import quadrotor_lmpc
    i = quadrotor_lmpc(ph, device=-1).info()
    assert (i["n_ref"], i["m_ref"], i["neq_ref"]) == ((ph + 1) * 16 + ph * 4, 2 * (ph + 1) * 16 + (ph + 1) * 12 + ph * 4 + ph + 1, (ph + 1) * 16)
    assert i["nz"] == 4 * ph and i["mg"] == 3 * ph


def _dense_condensed_from_oracle_builder(b, nf):
    """independent numpy condensing of the reference QP (variables eliminated by the dynamics rows)"""
    d = b.d; nx, nu, ph, na = d.nx, d.nu, d.ph, d.na
    A = b.ssA[:nx, :nx]; B = b.ssA[:nx, nx:]
    nz = nf * nu
    blk = [0] + [min(i, nf) - 1 for i in range(1, ph + 1)]
    S = [np.zeros((nx, nz))]
    for i in range(1, ph + 1):
        Si = A @ S[-1]
        Si[:, blk[i] * nu:(blk[i] + 1) * nu] += B
        S.append(Si)
    # z = Z w + z0: stack [x_i; v_i] and delta_i
    Z = np.zeros((d.nvar, nz))
    for i in range(1, ph + 1):
        Z[i * na:i * na + nx] = S[i]
        Z[i * na + nx:(i + 1) * na, blk[i] * nu:(blk[i] + 1) * nu] = np.eye(nu)
    for i in range(ph):
        r = (ph + 1) * na + i * nu
        Z[r:r + nu, blk[i + 1] * nu:(blk[i + 1] + 1) * nu] += np.eye(nu)
        if i > 0:
            Z[r:r + nu, blk[i] * nu:(blk[i] + 1) * nu] -= np.eye(nu)
    return Z.T @ b.P @ Z, Z


def test_condensed_hessian_matches_reference_qp():
    """H = Z' P Z where z = Z w + z0 parametrises the reference's equality constraints"""
    from libmpc_amd.workloads import quadrotor_lmpc
    ph = 10
    c = quadrotor_lmpc(ph, device=-1)
    dims = c.debug_get("dims")
    nz, mg, ldz, ldg, ldy = (int(v) for v in dims[:5])
    H = c.debug_get("H").reshape(nz, ldz)[:, :nz]
    b = LN.quadrotor_builder(ph)
    Href, Z = _dense_condensed_from_oracle_builder(b, nf=ph)
    assert np.allclose(H, Href, rtol=1e-10, atol=1e-9)
    # general rows = rows of the reference A applied to Z
    Gr = c.debug_get("Gr").reshape(ldg, ldz)[:mg, :nz]
    rows = c.debug_get("g_refrow").astype(int)[:mg]
    assert np.allclose(Gr, (b.A @ Z)[rows], atol=1e-12)
    # ADMM matrix inverse and dual Hessian are what they claim to be
    Kinv = c.debug_get("Kinv").reshape(nz, ldz)[:, :nz]
    rho_b = c.debug_get("rho_b")[:nz]; rho_g = c.debug_get("rho_g")[:mg]
    K = H + 1e-6 * np.eye(nz) + np.diag(rho_b) + Gr.T @ (rho_g[:, None] * Gr)
    assert np.allclose(K @ Kinv, np.eye(nz), atol=1e-8)
    Y = c.debug_get("Y").reshape(ldy, ldy)
    N = np.vstack([np.eye(nz), Gr])
    Yref = N @ np.linalg.solve(H, N.T)
    assert np.allclose(Y[:nz, :nz], Yref[:nz, :nz], rtol=1e-8, atol=1e-10)
    assert np.allclose(Y[ldz:ldz + mg, ldz:ldz + mg], Yref[nz:, nz:], rtol=1e-8, atol=1e-10)
    assert np.allclose(c.debug_get("lw")[:nz], 9.6 - 10.5916) and np.allclose(c.debug_get("uw")[:nz], 13 - 10.5916)


def test_move_blocking_dimensions():
    """delta-u is free for steps 0..ch inclusive (ProblemBuilder.hpp:782-793): ch+1 free moves"""
    from libmpc_amd import LMPC
    spec = random_lmpc_spec(3)
    c = configure_random(LMPC(*spec["dims"], device=-1), spec)
    nx, nu, ndu, ny, ph, ch = spec["dims"]
    dims = c.debug_get("dims")
    assert int(dims[5]) == min(ph, ch + 1) and int(dims[0]) == min(ph, ch + 1) * nu
    # same spec is accepted by the oracle front-end
    configure_random(OracleFrontEnd(*spec["dims"]), spec)


def test_nlmpc_create_validates_before_touching_a_device():
    """mpcx_nlmpc_create: argument errors are MPCX_E_INVALID, a missing GPU is MPCX_E_DEVICE -- never a silent CPU path"""
    import ctypes as C
    from libmpc_amd import _capi
    lib = _capi.lib()
    h = C.c_void_p()
    assert lib.mpcx_nlmpc_create(99, 10, 5, 0.1, None, 0, 0, C.byref(h)) == _capi.E_INVALID          # unknown model
    assert lib.mpcx_nlmpc_create(1, 5, 6, 0.1, None, 0, 0, C.byref(h)) == _capi.E_INVALID           # ch > ph
    assert lib.mpcx_nlmpc_create(1, 0, 0, 0.1, None, 0, 0, C.byref(h)) == _capi.E_INVALID
    bad = (C.c_double * 3)(1.0, 2.0, 3.0)
    assert lib.mpcx_nlmpc_create(2, 10, 10, 0.1, bad, 3, 0, C.byref(h)) == _capi.E_INVALID          # the UGV takes 9 parameters
    import torch
    if not torch.cuda.is_available():
        assert lib.mpcx_nlmpc_create(1, 10, 5, 0.1, None, 0, 0, C.byref(h)) == _capi.E_DEVICE
        assert b"HIP device" in lib.mpcx_last_error()
    p = _capi.NLParams()
    lib.mpcx_nlparams_default(C.byref(p))
    assert (p.maximum_iteration, p.relative_ftol, p.relative_xtol, p.hard_constraints, p.enable_warm_start) == (100, -1.0, -1.0, 1, 0)


def test_ctypes_structures_match_the_c_header(tmp_path):
    """the Python mirror of every struct in include/mpcx.h has the size and field offsets the C compiler gives it"""
    import ctypes as C
    import shutil
    import subprocess
    from libmpc_amd import _capi
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    pairs = {"mpcx_dims": _capi.Dims, "mpcx_lparams": _capi.LParams, "mpcx_lmpc_batch": _capi.Batch, "mpcx_lmpc_info": _capi.Info,
             "mpcx_nlmpc_dims": _capi.NlmpcDims, "mpcx_nlparams": _capi.NLParams, "mpcx_nlmpc_batch": _capi.NlmpcBatch,
             "mpcx_nlmpc_source": _capi.NlmpcSource}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "mpcx.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} %zu", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf(" %zu", offsetof({cname}, {fname}));')
        lines.append('  printf("\\n");')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"; exe = tmp_path / "layout"
    src.write_text("\n".join(lines))
    subprocess.check_call(["gcc", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).strip().splitlines()
    for line in out:
        parts = line.split()
        cls = pairs[parts[0]]
        assert int(parts[1]) == C.sizeof(cls), parts[0]
        offs = [getattr(cls, f).offset for f, _ in cls._fields_]
        assert [int(p) for p in parts[2:]] == offs, parts[0]


def test_reference_setters_do_not_rebuild_the_controller():
    """LMPC::setReferences / setExogenousInputs leave the time-invariant terms alone in the reference (LOptimizer.hpp:130-181 only
    stores the matrices); here they trigger a reference-only refresh, never the condensing"""
    import ctypes as C
    from libmpc_amd.workloads import quadrotor_lmpc
    c = quadrotor_lmpc(10, device=-1)
    c.info()
    full, refs = C.c_int(), C.c_int()
    lib = c._lib
    lib.mpcx_lmpc_debug_setup_counts(c._h, C.byref(full), C.byref(refs))
    assert (full.value, refs.value) == (1, 0)
    yref = np.zeros(12); yref[2] = 0.7
    for k in range(3):
        assert c.setReferences(yref * (k + 1), np.zeros(4), np.zeros(4), (0, 10))
        assert c.setExogenousInputs(np.zeros(4), (-1, -1))
        c.info()
    lib.mpcx_lmpc_debug_setup_counts(c._h, C.byref(full), C.byref(refs))
    assert (full.value, refs.value) == (1, 3)
    c.setInputBounds([-1.0] * 4, [1.0] * 4, (0, 10))          # a real change of the controller does rebuild
    c.info()
    lib.mpcx_lmpc_debug_setup_counts(c._h, C.byref(full), C.byref(refs))
    assert full.value == 2


@pytest.mark.parametrize("ph", [10, 20, 50])
def test_packed_mfma_operands_are_the_maps_rearranged(ph):
    """lmpc_pack_mfma_tiles (what lmpc_solve_group, lmpc_assemble_mfma and lmpc_cost_mfma read their A operands from): entry ((t G + g) 64 + lane) 4 + e of
    the packed copy is entry (row 16 t + lane % 16, column 4 (4 g + e) + lane // 16) of the map, zero beyond its columns -- every entry of the map exactly once"""
    from libmpc_amd.workloads import quadrotor_lmpc
    c = quadrotor_lmpc(ph, device=-1)
    kin, _, _, _, _, nz16, _, _, _, _, rowsA, ldy16 = c.debug_get("dims_maps").astype(int)
    for name, rows, K in (("MA0", rowsA, kin), ("MA1", rowsA, kin), ("Ym", ldy16, nz16)):
        src = c.debug_get(name).reshape(K, rows)            # column-major rows x K: [k][row]
        G = (K + 15) // 16
        p = c.debug_get(name + "p").reshape(rows // 16, G, 64, 4)
        lane = np.arange(64)
        for g in range(G):
            for e in range(4):
                k = 4 * (4 * g + e) + lane // 16
                want = np.where((k < K)[None, :], np.stack([src[np.minimum(k, K - 1), 16 * t + lane % 16] for t in range(rows // 16)]), 0.0)
                assert np.array_equal(p[:, g, :, e], want), (name, g, e)
        assert np.isclose(np.abs(p).sum(), np.abs(src).sum(), rtol=1e-12)
