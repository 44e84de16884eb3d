#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
T=${1:-r04c}
for w in ugv osc8 osc6; do ( MPCX_DEBUG_OCCUPANCY=1 timeout 300 python tools/nlmpc_phases.py $w 1024 ) > $O/${T}_phases_$w.txt 2>&1; grep -v amdgpu.ids $O/${T}_phases_$w.txt; done
for w in ugv osc8 osc6 vanderpol; do
  ( timeout 300 python bench.py --workload $w --cpu-seconds 0 ) > $O/${T}_bench_$w.json 2> $O/${T}_bench_$w.err; cut -c1-200 $O/${T}_bench_$w.json
done
( timeout 1200 python -m pytest tests/test_nlmpc_gpu.py tests/test_nlmpc_hooks.py -m gpu -q -rA --timeout 300 2>&1 | grep -v "^PASSED" ) > $O/${T}_pytest_nlmpc.log 2>&1; tail -5 $O/${T}_pytest_nlmpc.log | cut -c1-250; grep -B2 -A25 "^E  \|Error" $O/${T}_pytest_nlmpc.log | head -60
