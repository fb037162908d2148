#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp6
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_solve_driver.py -x -q -m gpu > gpurun_out/exp6/pytest_parity.log 2>&1; echo "rc $?" >> gpurun_out/exp6/pytest_parity.log; tail -3 gpurun_out/exp6/pytest_parity.log
timeout 1200 python -m pytest tests/test_gpu_ref_caller.py -x -q -m gpu -k "not 100" > gpurun_out/exp6/pytest_ref.log 2>&1; echo "rc $?" >> gpurun_out/exp6/pytest_ref.log; tail -3 gpurun_out/exp6/pytest_ref.log
export OPENBLAS_NUM_THREADS=1
for n in 60 100; do PASTIX_AMD_VERBOSE=1 timeout 600 oracle/_ref/ref_harness_d_ob_amd amd rlap3d $n llt 1 /dev/null 64 128 2>&1 | grep -E "one-shot|wall_sopalin" | tee -a gpurun_out/exp6/refcaller.txt; done
