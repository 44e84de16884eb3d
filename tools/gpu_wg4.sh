#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
T=${1:-r04e}
for w in ugv osc8; do ( MPCX_LIBRARY=$PWD/libmpc_amd/libmpcx_stats.so timeout 300 python tools/nlmpc_phases.py $w 1024 ) > $O/${T}_phases_$w.txt 2>&1; grep -v amdgpu.ids $O/${T}_phases_$w.txt; done
for w in ugv osc8 osc6 vanderpol; do
  ( timeout 300 python bench.py --workload $w --cpu-seconds 0 ) > $O/${T}_bench_$w.json 2> $O/${T}_bench_$w.err; cut -c1-200 $O/${T}_bench_$w.json
done
