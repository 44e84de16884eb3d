#!/bin/bash
# NLMPC workgroup form: folded blocks + reduced rows in LDS (MPCX_NLMPC_BLOCKS=1) against the workspace (=0, three workgroups per CU at four
# wavefronts per instance), at the batches of bench.py's workloads.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
line() { ( "$@" MPCX_DEBUG_OCCUPANCY=1 timeout 300 python bench.py --workload $w --cpu-seconds 0 --steps 3 --warmup 1 --nlmpc-extra 0 ) 2>&1 | grep -o '[0-9]* bytes of LDS each; resident per CU: [0-9]*\|"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"solved_fraction": [0-9.]*' | sort -u | tr '\n' ' '; }
for w in ugv osc6 osc8; do
  echo "$w wavefront form:                 $(line env MPCX_NLMPC_FORM=wave)"
  for bl in 1 0; do for wv in 4 2; do
    echo "$w workgroup form, $wv waves, blocks=$bl: $(line env MPCX_NLMPC_FORM=wg MPCX_NLMPC_WAVES=$wv MPCX_NLMPC_BLOCKS=$bl)"
  done; done
done
timeout 600 python -m pytest tests/test_nlmpc_forms.py -m gpu -x -q 2>&1 | tail -3
