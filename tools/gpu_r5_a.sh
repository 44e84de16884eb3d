#!/bin/bash
# round 5, first GPU call: the restructured workgroup form (fused dual step, wavefront-0 warm start, row-wise condensing, MFMA Schur complement,
# eight wavefronts for config 5) -- tests, the forms side by side, the phase profiles
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
T=${1:-r05a}
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/${T}_smoke.log 2>&1; echo "smoke rc $?"
( timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -25 ) > $O/${T}_pytest.log 2>&1; tail -3 $O/${T}_pytest.log | cut -c1-220
( timeout 600 python tools/nlmpc_variants.py ) > $O/${T}_variants.txt 2>&1; grep -v "amdgpu.ids\|Warn" $O/${T}_variants.txt
for w in ugv osc8; do ( MPCX_NLMPC_FORM=wg MPCX_LIBRARY=$PWD/libmpc_amd/libmpcx_stats.so timeout 300 python tools/nlmpc_phases.py $w 256 ) > $O/${T}_phases_wg_$w.txt 2>&1; grep -v "amdgpu.ids\|Warn" $O/${T}_phases_wg_$w.txt | tail -14; done
