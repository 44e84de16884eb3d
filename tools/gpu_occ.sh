#!/bin/bash
# per-CU delivery of the NLMPC kernel forms against the instances a CU holds (tools/nlmpc_occupancy.py); output
# profiles/rNN_nlmpc_occupancy.txt
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
w=${1:-ugv}
run() { timeout 300 python tools/nlmpc_occupancy.py "$@" 2>&1 | grep -v "amdgpu.ids\|Warn"; }
echo "== wavefront form (nlmpc_sqp)"; MPCX_NLMPC_FORM=wave run $w 1 2 4 8 16
echo "== workgroup form, 4 wavefronts, blocks and reduced rows in LDS"; MPCX_NLMPC_FORM=wg MPCX_NLMPC_WAVES=4 MPCX_NLMPC_BLOCKS=1 run $w 1 2 4
echo "== workgroup form, 4 wavefronts, blocks and reduced rows in the workspace"; MPCX_NLMPC_FORM=wg MPCX_NLMPC_WAVES=4 MPCX_NLMPC_BLOCKS=0 run $w 1 2 4 8
echo "== workgroup form, 2 wavefronts, workspace"; MPCX_NLMPC_FORM=wg MPCX_NLMPC_WAVES=2 MPCX_NLMPC_BLOCKS=0 run $w 1 2 4 8
echo "== workgroup form, 1 wavefront, workspace"; MPCX_NLMPC_FORM=wg MPCX_NLMPC_WAVES=1 MPCX_NLMPC_BLOCKS=0 run $w 1 2 4 8
