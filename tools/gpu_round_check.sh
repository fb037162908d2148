mkdir -p gpurun_out/r02a
python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a/pytest.log
tail -5 gpurun_out/r02a/pytest.log
python bench.py --steps 3 --warmup 1 > gpurun_out/r02a/bench200.json 2> gpurun_out/r02a/bench200.err
tail -c 1500 gpurun_out/r02a/bench200.json
python bench.py --grid 100 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02a/bench100.json 2>> gpurun_out/r02a/bench200.err
./tools/bench_update 4096 256 128 64 > gpurun_out/r02a/mb.txt
./tools/bench_update 8192 16 128 4096 >> gpurun_out/r02a/mb.txt
PM=64 PN=64 PDR=32 PDC=16 ./tools/bench_update 8192 12 64 64 >> gpurun_out/r02a/mb.txt
cat gpurun_out/r02a/mb.txt
python bench.py --gpus 2 --grid 40 > gpurun_out/r02a/bench_gpus2.txt 2>&1; echo "rc=$?" >> gpurun_out/r02a/bench_gpus2.txt; tail -3 gpurun_out/r02a/bench_gpus2.txt
