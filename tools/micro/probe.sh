#!/bin/bash
# both investigations of DESIGN.md section 9-2 in one GPU call; output: profiles/rNN_probe_*.txt
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
bash tools/micro/exec_mask.sh > gpurun_out/probe_exec_mask.txt 2>&1
TMPDIR=/tmp timeout 600 python tools/micro/blk_two_level.py 256 > gpurun_out/probe_blk_two_level.txt 2>&1
cat gpurun_out/probe_exec_mask.txt gpurun_out/probe_blk_two_level.txt
