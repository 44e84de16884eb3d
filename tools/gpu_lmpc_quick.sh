cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_lmpc_gpu.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4
for k in 0 1 2 3; do echo -n "cut $k: "; MPCX_LIBRARY=$PWD/libmpc_amd/libmpcx_cut$k.so python tools/group_cut.py 2>&1 | grep "step ms"; done; echo -n "full: "; python tools/group_cut.py 2>&1 | grep "step ms"
timeout 300 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --nlmpc-extra 0 2>/dev/null | cut -c1-300
