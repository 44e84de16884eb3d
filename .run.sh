cd $GRAFT_REPO_ROOT
timeout -k 5 120 python -m pytest tests/test_lmpc_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout -k 5 120 python bench.py --steps 50 --warmup 5 --cpu-seconds 0 2>&1 | tail -1 | cut -c1-300
