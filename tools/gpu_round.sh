#!/bin/bash
# One gpurun call of a round: smoke, the GPU test suite (full report, not -x), the profiles (kernel trace, HBM traffic, SQ / TCC
# counters), then the bench lines (which quote the traffic and counter summaries just measured on this very tree -- bench.py refuses
# a summary stamped with another digest of the kernel sources), and the phase profiles of the two solver kernels.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
T=${1:-r03}
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/${T}_smoke.log 2>&1; echo "smoke rc $?"
( timeout 1800 python -m pytest tests -m gpu -q -rA --timeout 600 2>&1 | grep -v "^PASSED" ) > $O/${T}_pytest.log 2>&1; tail -4 $O/${T}_pytest.log | cut -c1-220
timeout 300 tools/profile.sh $T lmpc20_b4096 --steps 60 --warmup 10
timeout 300 tools/profile.sh $T lmpc50_b32768 --config 4 --steps 10 --warmup 2
timeout 400 tools/profile.sh $T ugv_b4096 --workload ugv --steps 3 --warmup 1
timeout 400 tools/profile.sh $T osc8_b1024 --workload osc8 --steps 3 --warmup 1
timeout 300 tools/profile.sh $T lmpchetero20_b4096 --workload lmpc-hetero --steps 30 --warmup 3
timeout 500 tools/profile_sq.sh $T lmpc20_b4096 --steps 40 --warmup 5 > $O/${T}_sq_lmpc20.log 2>&1
timeout 500 tools/profile_sq.sh $T lmpc50_b32768 --config 4 --steps 6 --warmup 2 > $O/${T}_sq_lmpc50.log 2>&1
timeout 600 tools/profile_sq.sh $T ugv_b4096 --workload ugv --steps 2 --warmup 1 > $O/${T}_sq_ugv.log 2>&1
timeout 600 tools/profile_sq.sh $T osc8_b1024 --workload osc8 --steps 2 --warmup 1 > $O/${T}_sq_osc8.log 2>&1
timeout 500 tools/profile_sq.sh $T lmpchetero20_b4096 --workload lmpc-hetero --steps 20 --warmup 3 > $O/${T}_sq_hetero.log 2>&1
cp $O/${T}_pmc_traffic_*.json $O/${T}_sq_*.json profiles/ 2>/dev/null
( timeout 300 python bench.py --steps 200 --warmup 20 ) > $O/${T}_bench_lmpc20.json 2> $O/${T}_bench_lmpc20.err; cut -c1-400 $O/${T}_bench_lmpc20.json
( MPCX_FORCE_DIST=1 timeout 200 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --pipeline-streams 0 ) 2> $O/${T}_bench_lmpc20_rccl1.err | grep "^{" > $O/${T}_bench_lmpc20_rccl1.json; cut -c1-300 $O/${T}_bench_lmpc20_rccl1.json
( timeout 300 python bench.py --config 4 --steps 20 --warmup 3 --cpu-seconds 0 --pipeline-streams 0 ) > $O/${T}_bench_lmpc50.json 2> $O/${T}_bench_lmpc50.err; cut -c1-300 $O/${T}_bench_lmpc50.json
( timeout 300 python bench.py --workload lmpc-hetero --steps 50 --warmup 5 ) > $O/${T}_bench_lmpchetero.json 2> $O/${T}_bench_lmpchetero.err; cut -c1-300 $O/${T}_bench_lmpchetero.json
for w in vanderpol ugv osc6 osc8; do
  ( timeout 600 python bench.py --workload $w ) > $O/${T}_bench_$w.json 2> $O/${T}_bench_$w.err; cut -c1-330 $O/${T}_bench_$w.json; tail -2 $O/${T}_bench_$w.err | grep -v amdgpu.ids
done
for w in osc8 ugv; do ( MPCX_LIBRARY=$PWD/libmpc_amd/libmpcx_stats.so timeout 300 python tools/nlmpc_phases.py $w 1024 ) > $O/${T}_phases_$w.txt 2>&1; grep -v amdgpu.ids $O/${T}_phases_$w.txt; done
( for B in 64 4096; do MPCX_LIBRARY=$PWD/libmpc_amd/libmpcx_prof.so timeout 120 python tools/fast_phases.py 20 $B; done; timeout 120 python tools/group_phases.py 20 4096; for B in 4096 32768; do timeout 200 python tools/lmpc_ab.py $B 20; done ) 2>&1 | grep -v amdgpu.ids > $O/${T}_lmpc_phases_and_ab.txt; cat $O/${T}_lmpc_phases_and_ab.txt
