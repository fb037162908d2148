mkdir -p gpurun_out/r02j
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r02j/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02j/pytest.log
tail -15 gpurun_out/r02j/pytest.log
