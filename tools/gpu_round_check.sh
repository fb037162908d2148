export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_solve_driver.py -q -x 2>&1 | tail -2
rm -rf /tmp/le
rocprofv3 --kernel-trace --output-format csv -d /tmp/le -- python3 tools/dev_bench.py -n 100 --reps 2 > /tmp/le.log 2>&1
python3 tools/leaf_timeline.py /tmp/le 60 | grep -E "diag|busy"
for n in 60 100 160; do python tools/dev_bench.py -n $n --reps 4 2>&1 | tail -1 | cut -c1-75; PASTIX_AMD_NARROW_DIAG=0 python tools/dev_bench.py -n $n --reps 4 2>&1 | tail -1 | cut -c1-75; done
