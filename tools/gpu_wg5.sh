#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for w in ugv osc6 osc8; do for wv in 1 2; do echo "$w waves $wv: $( ( MPCX_NLMPC_WAVES=$wv MPCX_DEBUG_OCCUPANCY=1 timeout 300 python bench.py --workload $w --cpu-seconds 0 --steps 2 --warmup 1 ) 2>&1 | grep -o 'resident per CU: [0-9]*\|"value": [0-9.]*' | tr '\n' ' ')"; done; done
