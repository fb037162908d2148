#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp15
timeout 2400 python -m pytest tests -q -m gpu --durations=5 > gpurun_out/exp15/pytest_gpu.log 2>&1; echo "rc $?" >> gpurun_out/exp15/pytest_gpu.log; tail -12 gpurun_out/exp15/pytest_gpu.log
run() { name=$1; shift; timeout 1500 python tools/soak_run.py "$@" > gpurun_out/exp15/soak_$name.txt 2>&1; tail -1 gpurun_out/exp15/soak_$name.txt | cut -c1-260; }
run det_d40_llt  -n 40 --facto 0 --reps 400 --check 1
run det_d40_ldlt -n 40 --facto 1 --reps 400 --check 1
run det_d48_lu   -n 48 --facto 2 --reps 200 --check 1
run det_z20_ldlt -n 20 --facto 1 --complex --reps 200 --check 1
export OPENBLAS_NUM_THREADS=1
for rep in 1 2 3 4 5 6; do REF_ORDER_CONTIG=1 timeout 300 oracle/_ref/ref_harness_d_ob_amd cmp rlap3d 60 lu 32 /dev/null 2>/dev/null | grep '"cmp"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('cmp 60 lu: %.2e %d' % (max(d['rel_L'],d['rel_U']), d['worst_cblk']))"; done
