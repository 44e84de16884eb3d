#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for w in ugv; do
echo "== wave form"; MPCX_NLMPC_FORM=wave timeout 300 python tools/nlmpc_occupancy.py $w 1 2 4 8 16 2>&1 | grep -v Warn
echo "== wg/4 blocks in LDS"; MPCX_NLMPC_FORM=wg MPCX_NLMPC_WAVES=4 MPCX_NLMPC_BLOCKS=1 timeout 300 python tools/nlmpc_occupancy.py $w 1 2 4 2>&1 | grep -v Warn
echo "== wg/4 blocks in workspace"; MPCX_NLMPC_FORM=wg MPCX_NLMPC_WAVES=4 MPCX_NLMPC_BLOCKS=0 timeout 300 python tools/nlmpc_occupancy.py $w 1 2 3 6 2>&1 | grep -v Warn
echo "== wg/2 blocks in workspace"; MPCX_NLMPC_FORM=wg MPCX_NLMPC_WAVES=2 MPCX_NLMPC_BLOCKS=0 timeout 300 python tools/nlmpc_occupancy.py $w 1 2 3 6 2>&1 | grep -v Warn
echo "== wg/1 blocks in workspace"; MPCX_NLMPC_FORM=wg MPCX_NLMPC_WAVES=1 MPCX_NLMPC_BLOCKS=0 timeout 300 python tools/nlmpc_occupancy.py $w 1 2 3 6 2>&1 | grep -v Warn
done
