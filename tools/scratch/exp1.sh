#!/bin/bash
# scratch: first look at the one-kernel run (diagonal tasks as tickets)
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp1
timeout 900 python -m pytest tests/test_gpu_run_schedule.py -x -q -m gpu > gpurun_out/exp1/pytest_run.log 2>&1; echo "pytest rc $?" >> gpurun_out/exp1/pytest_run.log
tail -5 gpurun_out/exp1/pytest_run.log
for k in 1 0; do
  PASTIX_AMD_RUN_ONEK=$k timeout 600 python tools/dev_run_ab.py -n 60 100 --reps 5 > gpurun_out/exp1/ab_onek$k.log 2>&1
  cat gpurun_out/exp1/ab_onek$k.log
done
for k in 1 0; do
  PASTIX_AMD_RUN_ONEK=$k timeout 900 python tools/soak_run.py -n 60 --reps 5000 --tag onek$k > gpurun_out/exp1/soak60_onek$k.log 2>&1
  tail -3 gpurun_out/exp1/soak60_onek$k.log
done
