# A/B of two builds of the library on one box: tools/gpu_ab.sh libA.so libB.so  (names under libmpc_amd/)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
A=${1:-libmpcx_nobal.so}; B=${2:-libmpcx.so}
for rep in 1 2 3; do for lib in $A $B; do echo -n "$lib: "; MPCX_LIBRARY=$PWD/libmpc_amd/$lib python tools/group_cut.py 2>&1 | grep "step ms" | cut -c1-20; done; done
MPCX_LIBRARY=$PWD/libmpc_amd/$B python tools/group_phases.py 20 4096 2>&1 | grep -v "amdgpu.ids\|Warn" | tail -11
