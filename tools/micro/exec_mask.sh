#!/bin/bash
# DESIGN.md section 9-2, first finding: the wavefront form's phase boundaries carry a scheduling barrier since a build of round 2 reached the
# condensing phase of the 6-oscillator network with lanes missing from EXEC.  This runs the canary test, the EXEC-guard sweep and the
# whole NLMPC GPU suite on the probe build (make -C libmpc_amd/csrc probe: -DMPCX_NL_NO_LAP_BARRIER, the barrier removed) to see whether the
# failure is still there.  Output: profiles/rNN_probe_exec_mask.txt
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp MPCX_LIBRARY=$PWD/libmpc_amd/libmpcx_probe.so MPCX_NLMPC_FORM=wave
echo "== canary + NLMPC suite on the build without the barrier"
timeout 900 python -m pytest tests/test_nlmpc_gpu.py tests/test_nlmpc_forms.py -m gpu -q -x 2>&1 | grep -E 'passed|failed|error' | tail -4
echo "== shape sweep (EXEC guard) on the build without the barrier"
timeout 900 python tools/nlmpc_stress.py 2>&1 | grep -v 'amdgpu.ids' | tail -60
