#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp3
timeout 1200 python -m pytest tests/test_gpu_run_schedule.py -x -q -m gpu > gpurun_out/exp3/pytest_run.log 2>&1; echo "pytest rc $?" >> gpurun_out/exp3/pytest_run.log
tail -5 gpurun_out/exp3/pytest_run.log
for k in 1 0; do
  for cfg in "-n 60 --facto 2" "-n 100 --facto 2" "-n 60 --facto 1" "-n 100 --facto 1" "-n 32 --facto 1 --complex" "-n 48 --facto 1 --complex"; do
    PASTIX_AMD_RUN_ONEK=$k timeout 600 python tools/soak_run.py $cfg --reps 20 --tag onek$k 2>&1 | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('onek$k', d['n'], d['facto'], d['complex'], 'median %.3f ms  stops %d digests %d' % (d['median_ms'], d['stops'], d['distinct_digests']))" | tee -a gpurun_out/exp3/ab.log
  done
done
