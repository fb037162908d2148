export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_solve_driver.py tests/test_gpu_ref_caller.py -q -x 2>&1 | tail -2 | cut -c1-200
for n in 60 100 160; do python tools/dev_bench.py -n $n --reps 5 2>&1 | tail -1 | cut -c1-75; done
cd tools; for cfg in "128 1 128" "24 8192 300"; do ./bench_diag $cfg | tail -1; done
