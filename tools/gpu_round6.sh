#!/bin/bash
# One gpurun call of round 6: smoke, the GPU test suite, kernel traces / HBM traffic / SQ counters of the quoted workloads, then the bench lines
# (which quote the traffic and counter summaries just measured on this very tree: bench.py refuses a summary stamped with another digest of the
# kernel sources), the phase profiles and the side-by-side tables of the NLMPC kernel forms.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
T=${1:-r06}
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/${T}_smoke.log 2>&1; echo "smoke rc $?"
( timeout 1800 python -m pytest tests -m gpu -q -rA --timeout 600 2>&1 | grep -v "^PASSED" ) > $O/${T}_pytest.log 2>&1; tail -3 $O/${T}_pytest.log | cut -c1-220
timeout 300 tools/profile.sh $T lmpc20_b4096 --steps 60 --warmup 10 --nlmpc-extra 0
timeout 300 tools/profile.sh $T lmpc50_b32768 --config 4 --steps 10 --warmup 2 --nlmpc-extra 0
timeout 400 tools/profile.sh $T ugv_b4096 --workload ugv --steps 3 --warmup 1
timeout 400 tools/profile.sh $T osc8_b1024 --workload osc8 --steps 3 --warmup 1
timeout 300 tools/profile.sh $T ugv_b256 --workload ugv --batch 256 --steps 5 --warmup 1
timeout 300 tools/profile.sh $T osc8_b256 --workload osc8 --batch 256 --steps 3 --warmup 1
timeout 300 tools/profile.sh $T vanderpol_b4096 --workload vanderpol --steps 50 --warmup 5
timeout 300 tools/profile.sh $T lmpchetero20_b4096 --workload lmpc-hetero --steps 30 --warmup 3
timeout 600 tools/profile_sq.sh $T lmpc20_b4096 --steps 40 --warmup 5 --nlmpc-extra 0 > $O/${T}_sq_lmpc20.log 2>&1
timeout 900 tools/profile_sq.sh $T ugv_b4096 --workload ugv --steps 2 --warmup 1 > $O/${T}_sq_ugv4096.log 2>&1
timeout 900 tools/profile_sq.sh $T osc8_b1024 --workload osc8 --steps 2 --warmup 1 > $O/${T}_sq_osc81024.log 2>&1
timeout 600 tools/profile_sq.sh $T vanderpol_b4096 --workload vanderpol --steps 30 --warmup 3 > $O/${T}_sq_vdp.log 2>&1
timeout 600 tools/profile_sq.sh $T lmpc50_b32768 --config 4 --steps 6 --warmup 2 --nlmpc-extra 0 > $O/${T}_sq_lmpc50.log 2>&1
timeout 600 tools/profile_sq.sh $T lmpchetero20_b4096 --workload lmpc-hetero --steps 20 --warmup 3 > $O/${T}_sq_lmpchetero.log 2>&1
timeout 600 tools/profile_sq.sh $T ugv_b256 --workload ugv --batch 256 --steps 3 --warmup 1 > $O/${T}_sq_ugv256.log 2>&1
timeout 600 tools/profile_sq.sh $T osc8_b256 --workload osc8 --batch 256 --steps 3 --warmup 1 > $O/${T}_sq_osc8256.log 2>&1
cp $O/${T}_pmc_traffic_*.json $O/${T}_sq_*.json $O/${T}_kernel_trace_stats_*.txt profiles/ 2>/dev/null
( timeout 300 python bench.py --steps 200 --warmup 20 ) > $O/${T}_bench_lmpc20.json 2> $O/${T}_bench_lmpc20.err; cut -c1-300 $O/${T}_bench_lmpc20.json
( MPCX_FORCE_DIST=1 timeout 400 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --pipeline-streams 0 --nlmpc-extra 0 ) 2> $O/${T}_bench_lmpc20_rccl1.err | grep "^{" > $O/${T}_bench_lmpc20_rccl1.json; cut -c1-200 $O/${T}_bench_lmpc20_rccl1.json
( MPCX_FORCE_DIST=1 timeout 300 python bench.py --config 4 --steps 20 --warmup 3 --cpu-seconds 0 --pipeline-streams 0 ) 2> $O/${T}_bench_lmpc50_rccl1.err | grep "^{" > $O/${T}_bench_lmpc50_rccl1.json; cut -c1-200 $O/${T}_bench_lmpc50_rccl1.json
( MPCX_FORCE_DIST=1 timeout 300 python bench.py --config 5 --cpu-seconds 0 ) 2> $O/${T}_bench_osc8_rccl1.err | grep "^{" > $O/${T}_bench_osc8_rccl1.json; cut -c1-200 $O/${T}_bench_osc8_rccl1.json
( timeout 300 python bench.py --config 4 --steps 20 --warmup 3 --cpu-seconds 0 --pipeline-streams 0 ) > $O/${T}_bench_lmpc50.json 2> $O/${T}_bench_lmpc50.err; cut -c1-200 $O/${T}_bench_lmpc50.json
( timeout 300 python bench.py --workload lmpc-hetero --steps 50 --warmup 5 ) > $O/${T}_bench_lmpchetero.json 2> $O/${T}_bench_lmpchetero.err; cut -c1-200 $O/${T}_bench_lmpchetero.json
for w in vanderpol ugv osc6 osc8; do
  ( timeout 600 python bench.py --workload $w ) > $O/${T}_bench_$w.json 2> $O/${T}_bench_$w.err; cut -c1-200 $O/${T}_bench_$w.json
done
for wb in "ugv 256" "osc8 256"; do set -- $wb
  ( timeout 300 python bench.py --workload $1 --batch $2 --cpu-seconds 0 --steps 5 --warmup 1 ) > $O/${T}_bench_$1_b$2.json 2> $O/${T}_bench_$1_b$2.err; cut -c1-200 $O/${T}_bench_$1_b$2.json
done
# lmpc_solve_group cut short (tools/group_cut.sh built libmpcx_cut{0..3}.so before this call): what each part of its assemble phase costs a launch
if [ -f libmpc_amd/libmpcx_cut0.so ]; then
  ( for k in 0 1 2 3; do echo -n "kernel returns after part $k (0 entry, 1 inputs staged, 2 first product, 3 second product): "; MPCX_LIBRARY=$PWD/libmpc_amd/libmpcx_cut$k.so timeout 120 python tools/group_cut.py 2>&1 | grep "step ms"; done
    echo -n "the whole kernel: "; timeout 120 python tools/group_cut.py 2>&1 | grep "step ms" ) > $O/${T}_group_cut.txt 2>&1; cat $O/${T}_group_cut.txt
fi
( timeout 200 python tools/group_phases.py 20 4096 2>&1 | grep -v "amdgpu.ids\|Warn" ) > $O/${T}_group_phases_lmpc20_b4096.txt; tail -12 $O/${T}_group_phases_lmpc20_b4096.txt
for wb in "osc8 256" "osc8 1024" "ugv 256" "ugv 4096" "osc6 1024"; do set -- $wb
  ( MPCX_LIBRARY=$PWD/libmpc_amd/libmpcx_stats.so timeout 300 python tools/nlmpc_phases.py $1 $2 ) > $O/${T}_phases_wg_$1_b$2.txt 2>&1; grep -v "amdgpu.ids\|Warn" $O/${T}_phases_wg_$1_b$2.txt | tail -14
done
( timeout 300 python tools/nlmpc_closed_loop.py; timeout 300 python tools/nlmpc_closed_loop.py 256 20 ) > $O/${T}_nlmpc_closed_loop.json 2>&1; tail -6 $O/${T}_nlmpc_closed_loop.json | cut -c1-300
# the kernel forms side by side, and the time of one batched solve against the batch (one process: the overrides are read per handle)
SPECS=""
for w in ugv osc8 osc6; do for B in 1 64 256 512 1024; do SPECS="$SPECS $w:$B:default $w:$B:wave"; done; done
( timeout 900 python tools/nlmpc_variants.py ugv:4096:default ugv:4096:wg:4:1 ugv:4096:wg:2 ugv:4096:wave ugv:8192:default osc8:1024:default osc8:1024:wg:4 osc8:1024:wave osc8:2048:default \
    osc6:1024:default osc6:1024:wave vanderpol:4096:default vanderpol:4096:wave $SPECS ) 2>&1 | grep -v "amdgpu.ids\|Warn" > $O/${T}_nlmpc_forms.txt; head -14 $O/${T}_nlmpc_forms.txt
bash tools/gpu_occ.sh ugv > $O/${T}_nlmpc_occupancy.txt 2>&1
timeout 900 python tests/hunt_inconsistent_gpu.py 1024 2>&1 | grep -v "amdgpu.ids\|Warn\|wrapped\|fx =\|g = app" > $O/${T}_inconsistent_hunt.txt; tail -3 $O/${T}_inconsistent_hunt.txt
# round 6: the oscillator controllers with tight input bounds under every combination of the inverse-form switches; the UGV with the Gauss-Newton matrix installed before iteration K
timeout 900 python tools/osc_bounds_sweep.py 1024 > $O/${T}_osc_bounds_sweep.txt 2>&1; tail -8 $O/${T}_osc_bounds_sweep.txt | cut -c1-200
timeout 900 python tools/ugv_curv_sweep.py 0 5 8 10 12 14 16 20 30 > $O/${T}_ugv_curv_sweep.txt 2>&1; tail -4 $O/${T}_ugv_curv_sweep.txt | cut -c1-200
