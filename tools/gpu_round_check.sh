mkdir -p gpurun_out/r02q
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -q -x > gpurun_out/r02q/pytest_s4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02q/pytest_s4.log
tail -4 gpurun_out/r02q/pytest_s4.log
( for b in bench_update bench_update_s4; do
echo $b
./tools/$b 8192 16 128 64
./tools/$b 4096 256 128 64
./tools/$b 8192 16 128 4096
STRUCT=1 ROWS=40960 ./tools/$b 16384 16 128 16
done ) 2>&1 | tee gpurun_out/r02q/mb_s4.txt
for s in 66 196; do
PASTIX_AMD_DUMP_SLOT=$s:/tmp/slot$s.bin python tools/dev_bench.py -n 160 --reps 2 2>&1 | grep -E "dumped|rep 1"
./tools/replay_slot /tmp/slot$s.bin | grep "order 0"
./tools/replay_slot_s4 /tmp/slot$s.bin | grep "order 0"
done 2>&1 | tee gpurun_out/r02q/replay_s4.txt
python tools/dev_bench.py -n 100 --reps 3 2>&1 | tail -1
python tools/soak_determinism.py -n 40 --reps 10 2>&1 | tail -1
