#!/bin/bash
# NLMPC: the two forms of the solve kernel side by side (profiles/rNN_nlmpc_forms.txt): solves/s of bench.py's four NLMPC workloads with the
# default choice, the workgroup form forced (4, 2, 1 wavefronts per instance; at a batch beyond what is resident the plan takes the variant with
# the blocks in the workspace where that holds one more workgroup per CU) and the wavefront form forced.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
line() { ( "$@" timeout 300 python bench.py --workload $w --cpu-seconds 0 --steps $s --warmup 1 ) 2>&1 | grep -o 'resident per CU: [0-9]*\|"value": [0-9.]*\|"ms_per_step": [0-9.]*' | sort -u | tr '\n' ' '; }
for w in vanderpol ugv osc6 osc8; do
  s=3; [ $w = vanderpol ] && s=50
  echo "$w default:        $(line env MPCX_DEBUG_OCCUPANCY=1)"
  for wv in 4 2 1; do echo "$w workgroup form, $wv wavefront(s) per instance: $(line env MPCX_NLMPC_FORM=wg MPCX_NLMPC_WAVES=$wv MPCX_DEBUG_OCCUPANCY=1)"; done
  [ $w = ugv ] && echo "$w workgroup form, 4 wavefronts, blocks and reduced rows in LDS (two per CU): $(line env MPCX_NLMPC_FORM=wg MPCX_NLMPC_WAVES=4 MPCX_NLMPC_BLOCKS=1 MPCX_DEBUG_OCCUPANCY=1)"
  echo "$w wavefront form: $(line env MPCX_NLMPC_FORM=wave)"
done
