#!/bin/bash
# A short gpurun call: smoke, GPU tests (full report), the bench lines, SQP phase shares.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
T=${1:-r02}
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/${T}_smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/${T}_smoke.log | cut -c1-200
( timeout 1500 python -m pytest tests -m gpu -q -rA --timeout 600 2>&1 | grep -v "^PASSED" ) > $O/${T}_pytest.log 2>&1; tail -12 $O/${T}_pytest.log | cut -c1-220
( timeout 300 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 ) > $O/${T}_bench_lmpc20.json 2> $O/${T}_bench_lmpc20.err; cut -c1-330 $O/${T}_bench_lmpc20.json
for w in vanderpol ugv osc6 osc8; do
  ( timeout 400 python bench.py --workload $w --cpu-seconds 0 ) > $O/${T}_bench_$w.json 2> $O/${T}_bench_$w.err; cut -c1-200 $O/${T}_bench_$w.json; tail -2 $O/${T}_bench_$w.err | grep -v amdgpu.ids
done
for w in osc8 ugv; do ( MPCX_LIBRARY=$PWD/libmpc_amd/libmpcx_stats.so timeout 300 python tools/nlmpc_phases.py $w 256 ) > $O/${T}_phases_$w.txt 2>&1; cat $O/${T}_phases_$w.txt | grep -v amdgpu.ids; done
