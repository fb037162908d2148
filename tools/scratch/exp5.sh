#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp5
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -x -q -m gpu > gpurun_out/exp5/pytest_parity.log 2>&1; echo "rc $?" >> gpurun_out/exp5/pytest_parity.log; tail -3 gpurun_out/exp5/pytest_parity.log
timeout 2400 python -m pytest tests/test_gpu_ref_caller.py -x -q -m gpu --durations=8 > gpurun_out/exp5/pytest_ref.log 2>&1; echo "rc $?" >> gpurun_out/exp5/pytest_ref.log; tail -14 gpurun_out/exp5/pytest_ref.log
timeout 600 python tools/one_shot_timing.py 100 > gpurun_out/exp5/one_shot_100.json 2> gpurun_out/exp5/one_shot_100.err; cat gpurun_out/exp5/one_shot_100.json; tail -3 gpurun_out/exp5/one_shot_100.err
export OPENBLAS_NUM_THREADS=1
for n in 60 100; do PASTIX_AMD_VERBOSE=1 timeout 600 oracle/_ref/ref_harness_d_ob_amd amd rlap3d $n llt 1 /dev/null 64 128 2>&1 | grep -E "one-shot|wall_sopalin" | tee -a gpurun_out/exp5/refcaller.txt; done
REF_ORDER_CONTIG=1 timeout 900 oracle/_ref/ref_harness_d_ob_amd cmp rlap3d 100 llt 32 /dev/null 64 128 2>/dev/null | grep '"cmp"' | tee -a gpurun_out/exp5/refcaller.txt
