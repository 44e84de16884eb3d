#!/bin/bash
# first contact of the workgroup form with the GPU: NLMPC tests, then the four NLMPC bench lines in both forms
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
T=${1:-r04a}
( MPCX_DEBUG_OCCUPANCY=1 timeout 200 python bench.py --workload ugv --cpu-seconds 0 --steps 2 --warmup 1 ) > $O/${T}_first_ugv.json 2> $O/${T}_first_ugv.err; cut -c1-300 $O/${T}_first_ugv.json; grep nlmpc_sqp $O/${T}_first_ugv.err | head -3
( timeout 900 python -m pytest tests/test_nlmpc_gpu.py tests/test_nlmpc_hooks.py -m gpu -q -rA --timeout 300 -x 2>&1 | grep -v "^PASSED" ) > $O/${T}_pytest_nlmpc.log 2>&1; tail -15 $O/${T}_pytest_nlmpc.log | cut -c1-250
for w in ugv osc8 osc6 vanderpol; do
  ( MPCX_DEBUG_OCCUPANCY=1 timeout 300 python bench.py --workload $w --cpu-seconds 0 ) > $O/${T}_bench_$w.json 2> $O/${T}_bench_$w.err; cut -c1-260 $O/${T}_bench_$w.json; grep "nlmpc_sqp" $O/${T}_bench_$w.err | head -1
  ( MPCX_NLMPC_FORM=wave timeout 300 python bench.py --workload $w --cpu-seconds 0 ) > $O/${T}_bench_${w}_wave.json 2> $O/${T}_bench_${w}_wave.err; cut -c1-260 $O/${T}_bench_${w}_wave.json
done
for wv in 1 2; do ( MPCX_NLMPC_WAVES=$wv timeout 300 python bench.py --workload ugv --cpu-seconds 0 ) 2>/dev/null | cut -c1-200; done
for wv in 2 4; do ( MPCX_NLMPC_WAVES=$wv timeout 300 python bench.py --workload vanderpol --cpu-seconds 0 ) 2>/dev/null | cut -c1-200; done
