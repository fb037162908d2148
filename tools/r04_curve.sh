cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python tools/launch_curve.py gpurun_out/r04/launch_rate_curve.json 60 80 100 130 160 200 > gpurun_out/r04/launch_curve.log 2>&1
tail -c 1500 gpurun_out/r04/launch_curve.log
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "config2" 2>&1 | tail -3
