# quick look after a change to the LMPC kernels: the LMPC GPU tests, the headline line, config 4's line
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_lmpc_gpu.py tests/test_lmpc_hetero.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4
timeout 300 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --nlmpc-extra 0 2>/dev/null | cut -c100-330
timeout 300 python bench.py --config 4 --steps 20 --warmup 3 --cpu-seconds 0 --pipeline-streams 0 2>/dev/null | cut -c100-330
