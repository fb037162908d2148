mkdir -p gpurun_out/r02f
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02f/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02f/pytest.log
tail -6 gpurun_out/r02f/pytest.log
( PM=64 PN=64 PDR=32 PDC=16 ./tools/bench_update 8192 12 64 64
PM=64 PN=64 PDR=33 PDC=17 ./tools/bench_update 8192 12 64 64
PM=64 PN=64 PDR=32 PDC=16 ./tools/bench_update 8192 12 128 64
PM=100 PN=128 PDR=7 PDC=0 ./tools/bench_update 8192 12 128 64
PM=128 PN=40 PDR=0 PDC=50 ./tools/bench_update 8192 12 128 64
./tools/bench_update 8192 16 128 64
./tools/bench_update 4096 256 128 64 ) > gpurun_out/r02f/mb.txt 2>&1
cat gpurun_out/r02f/mb.txt
python tools/dev_bench.py -n 100 --reps 4 2>&1 | tail -3
python tools/dev_bench.py -n 60 --reps 4 2>&1 | tail -2
python tools/dev_bench.py -n 160 --reps 3 2>&1 | tail -2
