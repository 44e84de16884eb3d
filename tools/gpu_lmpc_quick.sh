cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_lmpc_gpu.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4
timeout 300 python bench.py --config 4 --steps 20 --warmup 3 --cpu-seconds 0 --pipeline-streams 0 2>/dev/null | cut -c1-400
rm -rf /tmp/prof50; rocprofv3 --kernel-trace --stats -d /tmp/prof50 -- python bench.py --config 4 --steps 10 --warmup 2 --nlmpc-extra 0 --cpu-seconds 0 --pipeline-streams 0 > /dev/null 2>&1
DB=$(find /tmp/prof50 -name "*.db" | head -1); python tools/rocprof_summary.py "$DB" /tmp/prof50/s.txt > /dev/null; head -8 /tmp/prof50/s.txt
