export TMPDIR=/tmp
PASTIX_AMD_QUAD_MIN=1 PASTIX_AMD_QUAD_FILL=2.0 timeout 1200 python -m pytest tests/test_gpu_dist.py tests/test_gpu_solve_driver.py tests/test_gpu_edges.py tests/test_gpu_ref_caller.py -q 2>&1 | tail -3 | cut -c1-200
PASTIX_AMD_QUAD_MIN=1 timeout 600 python tools/dev_bench_dist_local.py 60 4 2>&1 | tail -1
