#!/bin/bash
# One gpurun call of a round: smoke, the GPU test suite (full report, not -x), the profiles, then the bench lines (which quote
# the HBM traffic from the PMC summaries just measured on this very tree -- bench.py refuses a summary stamped with another
# digest of the kernel sources), and the SQP phase shares.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
T=${1:-r02}
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/${T}_smoke.log 2>&1; echo "smoke rc $?"
( timeout 1800 python -m pytest tests -m gpu -q -rA --timeout 600 2>&1 | grep -v "^PASSED" ) > $O/${T}_pytest.log 2>&1; tail -4 $O/${T}_pytest.log | cut -c1-220
timeout 400 tools/profile.sh $T lmpc20_b4096 --steps 60 --warmup 10
timeout 400 tools/profile.sh $T lmpc50_b32768 --config 4 --steps 10 --warmup 2
timeout 500 tools/profile.sh $T ugv_b4096 --workload ugv --steps 3 --warmup 1
timeout 500 tools/profile.sh $T osc8_b1024 --workload osc8 --steps 3 --warmup 1
cp $O/${T}_pmc_traffic_*.json profiles/ 2>/dev/null
( timeout 300 python bench.py --steps 200 --warmup 20 ) > $O/${T}_bench_lmpc20.json 2> $O/${T}_bench_lmpc20.err; cut -c1-400 $O/${T}_bench_lmpc20.json
( MPCX_FORCE_DIST=1 timeout 200 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --pipeline-streams 0 ) 2> $O/${T}_bench_lmpc20_rccl1.err | grep "^{" > $O/${T}_bench_lmpc20_rccl1.json; cut -c1-300 $O/${T}_bench_lmpc20_rccl1.json
( timeout 300 python bench.py --config 4 --steps 20 --warmup 3 --cpu-seconds 0 --pipeline-streams 0 ) > $O/${T}_bench_lmpc50.json 2> $O/${T}_bench_lmpc50.err; cut -c1-300 $O/${T}_bench_lmpc50.json
for w in vanderpol ugv osc6 osc8; do
  ( timeout 600 python bench.py --workload $w ) > $O/${T}_bench_$w.json 2> $O/${T}_bench_$w.err; cut -c1-330 $O/${T}_bench_$w.json; tail -2 $O/${T}_bench_$w.err | grep -v amdgpu.ids
done
for w in osc8 ugv; do ( MPCX_LIBRARY=$PWD/libmpc_amd/libmpcx_stats.so timeout 300 python tools/nlmpc_phases.py $w 1024 ) > $O/${T}_phases_$w.txt 2>&1; grep -v amdgpu.ids $O/${T}_phases_$w.txt; done
