#!/bin/bash
# NLMPC: time of one batched solve against the batch size, both forms of the kernel (profiles/rNN_nlmpc_latency.txt).  Up to the number of
# instances a form holds resident at once the time is that of one solve: the latency a controller sees.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
ms() { ( "$@" timeout 300 python bench.py --workload $w --batch $B --cpu-seconds 0 --steps 3 --warmup 1 ) 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2 | cut -c1-8; }
for w in ugv osc8 osc6 vanderpol; do
  for B in 1 64 256 512 1024 2048; do
    echo "$w batch $B: default $(ms env) ms   workgroup form $(ms env MPCX_NLMPC_FORM=wg) ms   wavefront form $(ms env MPCX_NLMPC_FORM=wave) ms"
  done
done
