#!/bin/bash
# builds libmpcx_cut{0..3}.so (see tools/group_cut.py) -- run here, the libraries travel with gpurun; not product builds
cd "$(dirname "$0")/../libmpc_amd/csrc"
for k in 0 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -I../../include -DMPCX_GROUP_CUT=$k -c -o build/lmpc_fast_cut$k.o -x hip lmpc_fast.hip &
done; wait
for k in 0 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libmpcx_cut$k.so $(ls build/*.o | grep -v "lmpc_fast\|_stats\|_probe") build/lmpc_fast_cut$k.o -ldl
done
