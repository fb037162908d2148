mkdir -p gpurun_out/r02o
for i in 1 2 3; do
timeout 1200 python -m pytest tests/test_gpu_dist.py tests/test_gpu_solve_driver.py -q > gpurun_out/r02o/pytest$i.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02o/pytest$i.log
tail -3 gpurun_out/r02o/pytest$i.log
done
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r02o/pytest_all.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02o/pytest_all.log
tail -3 gpurun_out/r02o/pytest_all.log
