#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for cap in 0 1792 1536 1280 1152 1024; do
  if [ $cap = 0 ]; then e=""; else e="MPCX_DEBUG_LDS_CAP=$cap"; fi
  echo "ugv wave form, LDS cap $cap: $( ( env $e MPCX_NLMPC_FORM=wave MPCX_DEBUG_OCCUPANCY=1 timeout 300 python bench.py --workload ugv --cpu-seconds 0 --steps 3 --warmup 1 ) 2>&1 | grep -o 'resident per CU: [0-9]*; factor rows in LDS [0-9]* of [0-9]*\|"value": [0-9.]*\|"solved_fraction": [0-9.]*' | sort -u | tr '\n' ' ')"
done
