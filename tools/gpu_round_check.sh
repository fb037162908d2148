mkdir -p gpurun_out/r02i
for o in 0 3; do
PASTIX_AMD_TASK_ORDER=$o python tools/dev_bench.py -n 160 --reps 3 > gpurun_out/r02i/t160_order$o.txt 2>&1
tail -2 gpurun_out/r02i/t160_order$o.txt
PASTIX_AMD_TASK_ORDER=$o python tools/dev_bench.py -n 100 --reps 4 > gpurun_out/r02i/t100_order$o.txt 2>&1
tail -2 gpurun_out/r02i/t100_order$o.txt
done
PASTIX_AMD_TASK_ORDER=3 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
