# quick look after a change to the NLMPC kernels: the NLMPC GPU tests, then the four NLMPC lines
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_nlmpc_gpu.py tests/test_nlmpc_forms.py tests/test_nlmpc_bounds_gpu.py tests/test_nlmpc_hooks.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
for w in vanderpol ugv osc6 osc8; do timeout 400 python bench.py --workload $w --cpu-seconds 0 2>/dev/null | cut -c1-190; done
