#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
T=${1:-r04b}
for w in ugv osc8 osc6; do ( timeout 300 python tools/nlmpc_phases.py $w 1024 ) > $O/${T}_phases_$w.txt 2>&1; grep -v amdgpu.ids $O/${T}_phases_$w.txt; done
( timeout 300 python tools/nlmpc_phases.py ugv 256 ) 2>&1 | grep -v amdgpu.ids
( MPCX_NLMPC_WAVES=1 timeout 300 python tools/nlmpc_phases.py ugv 1024 ) 2>&1 | grep -v amdgpu.ids
( timeout 1200 python -m pytest tests/test_nlmpc_gpu.py tests/test_nlmpc_hooks.py -m gpu -q -rA --timeout 300 2>&1 | grep -v "^PASSED" ) > $O/${T}_pytest_nlmpc.log 2>&1; tail -25 $O/${T}_pytest_nlmpc.log | cut -c1-250
