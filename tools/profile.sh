#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel trace + the two PMC passes (FETCH_SIZE and WRITE_SIZE do not fit one pass;
# counters are collected without any trace domain) of one bench.py workload; summaries land in gpurun_out/ under the
# names they are committed with in profiles/.
#   tools/profile.sh <round-tag> <workload-tag> <bench args...>       e.g.  tools/profile.sh r02 lmpc20_b4096 --steps 60
set -u
TAG=$1; WL=$2; shift 2
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
rm -rf /tmp/prof_$WL /tmp/pmc_f_$WL /tmp/pmc_w_$WL
rocprofv3 --kernel-trace --stats -d /tmp/prof_$WL -- python bench.py "$@" --cpu-seconds 0 --pipeline-streams 0 > $OUT/${TAG}_prof_${WL}.log 2>&1
DB=$(find /tmp/prof_$WL -name "*.db" | head -1)
if [ -n "$DB" ]; then python tools/rocprof_summary.py "$DB" $OUT/${TAG}_kernel_trace_stats_${WL}.txt > /dev/null; else
  STATS=$(find /tmp/prof_$WL -name "*kernel_stats.csv" | head -1); [ -n "$STATS" ] && cp "$STATS" $OUT/${TAG}_kernel_trace_stats_${WL}.csv; fi
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f_$WL --output-format csv -- python bench.py "$@" --cpu-seconds 0 --pipeline-streams 0 > $OUT/${TAG}_pmc_f_${WL}.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w_$WL --output-format csv -- python bench.py "$@" --cpu-seconds 0 --pipeline-streams 0 > $OUT/${TAG}_pmc_w_${WL}.log 2>&1
python tools/pmc_summary.py /tmp/pmc_f_$WL /tmp/pmc_w_$WL > $OUT/${TAG}_pmc_traffic_${WL}.json
tail -2 $OUT/${TAG}_prof_${WL}.log | cut -c1-600
