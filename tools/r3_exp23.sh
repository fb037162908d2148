#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_solve_driver.py tests/test_gpu_parity.py tests/test_gpu_configs_fullsize.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/dev_bench_solve.py -n 100 --reps 3 2>&1 | tail -2
timeout 300 python tools/dev_bench_solve.py -n 100 --facto 2 --reps 3 2>&1 | tail -1
timeout 600 python tools/dev_bench_solve.py -n 200 --reps 4 2>&1 | tail -3
