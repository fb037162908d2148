#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp8
timeout 1200 python -m pytest tests/test_gpu_run_schedule.py tests/test_gpu_solve_driver.py -q -m gpu > gpurun_out/exp8/pytest_a.log 2>&1; tail -12 gpurun_out/exp8/pytest_a.log
export OPENBLAS_NUM_THREADS=1 REF_ORDER_CONTIG=1
for r in 1 0; do for a in "60 lu" "40 lu" "60 llt"; do set -- $a; echo "RUN=$r $a"; PASTIX_AMD_RUN=$r timeout 300 oracle/_ref/ref_harness_d_ob_amd cmp rlap3d $1 $2 32 /dev/null 2>/dev/null | grep '"cmp"' | cut -c1-330; done; done
