cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for rep in 1 2 3; do for lib in libmpcx_base.so libmpcx.so; do echo -n "$lib: "; MPCX_LIBRARY=$PWD/libmpc_amd/$lib python tools/group_cut.py 2>&1 | grep "step ms" | cut -c1-20; done; done
python tools/group_phases.py 20 4096 2>&1 | grep -v "amdgpu.ids\|Warn" | tail -10
