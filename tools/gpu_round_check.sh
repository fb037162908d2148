export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_solve_driver.py tests/test_gpu_dist.py -q -x -k "fake or fill_matrix or at_scale" 2>&1 | tail -5
