#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp7
timeout 2400 python -m pytest tests -x -q -m gpu --durations=10 > gpurun_out/exp7/pytest_gpu.log 2>&1; echo "rc $?" >> gpurun_out/exp7/pytest_gpu.log; tail -16 gpurun_out/exp7/pytest_gpu.log
export OPENBLAS_NUM_THREADS=1
( echo "## lexicographic separators (fragmented bloks), gathered pieces on (default) / off (PASTIX_AMD_GATHER... via harness arg none: library default)"
for n in 60 80 100; do timeout 600 oracle/_ref/ref_harness_d_ob_amd amd rlap3d $n llt 1 /dev/null 64 128 2>/dev/null | tail -1; done
echo "## contiguous separators"
for n in 60 80 100; do REF_ORDER_CONTIG=1 timeout 600 oracle/_ref/ref_harness_d_ob_amd amd rlap3d $n llt 1 /dev/null 64 128 2>/dev/null | tail -1; done
echo "## own layouts"
for n in 60 80 100; do timeout 200 python tools/dev_bench.py -n $n --reps 3 2>/dev/null | tail -1; done ) > gpurun_out/exp7/refcaller.txt 2>&1
cat gpurun_out/exp7/refcaller.txt | cut -c1-400
