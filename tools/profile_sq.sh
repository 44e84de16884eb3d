#!/bin/bash
# Runs on the GPU box (gpurun): hardware-counter passes of one bench.py workload -- wave / busy / wait cycles, instruction mix,
# per-unit busy cycles (VALU, LDS, VMEM, scalar, MFMA), LDS bank conflicts, L2 hits / misses.  Each pass is its own rocprofv3 run
# with --kernel-trace only (counters and trace domains are never mixed); per-kernel means land in gpurun_out/<tag>_sq_<wl>.json,
# from where they are copied to profiles/.
#   tools/profile_sq.sh <round-tag> <workload-tag> <bench args...>       e.g.  tools/profile_sq.sh r03 lmpc20_b4096 --steps 40
set -u
TAG=$1; WL=$2; shift 2
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"
 "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
 "TCC_HIT TCC_MISS"
 "TCC_REQ TCC_TAG_STALL"
 "SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_IFETCH"
)
DIRS=""
i=0
for P in "${PASSES[@]}"; do
  D=/tmp/sq_${WL}_$i; rm -rf $D
  # a counter set the hardware cannot schedule makes rocprofv3 abort and then hang in its signal handler: bound every pass
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $P -d $D --output-format csv -- python bench.py "$@" --cpu-seconds 0 --pipeline-streams 0 > $OUT/${TAG}_sq_${WL}_pass$i.log 2>&1 || echo "pass $i failed: $P"
  DIRS="$DIRS $D"; i=$((i+1))
  python tools/pmc_summary.py $DIRS > $OUT/${TAG}_sq_${WL}.json 2>/dev/null
done
python - <<PY
import json
d = json.load(open("$OUT/${TAG}_sq_${WL}.json"))
ks = sorted({k for v in d.values() if isinstance(v, dict) for k in v})
for k in ks:
    print(k, {c: round(v[k]["mean"]) for c, v in d.items() if isinstance(v, dict) and k in v})
PY
