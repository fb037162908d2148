mkdir -p gpurun_out/r02b
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_parity.py -x -q > gpurun_out/r02b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02b/pytest.log
tail -15 gpurun_out/r02b/pytest.log
timeout 200 python tools/probe_rccl_same_gpu.py > gpurun_out/r02b/rccl_probe.txt 2>&1; tail -5 gpurun_out/r02b/rccl_probe.txt
timeout 600 python tools/dev_bench_dist_local.py 60 4 > gpurun_out/r02b/dist_local.txt 2>&1
timeout 600 python tools/dev_bench_dist_local.py 100 4 >> gpurun_out/r02b/dist_local.txt 2>&1
timeout 600 python tools/dev_bench_dist_local.py 100 8 >> gpurun_out/r02b/dist_local.txt 2>&1
cat gpurun_out/r02b/dist_local.txt
