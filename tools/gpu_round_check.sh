mkdir -p gpurun_out/r02n
bash tools/profile_round.sh 200 > gpurun_out/r02n/p200.log 2>&1
bash tools/profile_round.sh 100 > gpurun_out/r02n/p100.log 2>&1
NO_PMC=1 bash tools/profile_round.sh 48 --workload elasticity > gpurun_out/r02n/pz48.log 2>&1
python bench.py --workload elasticity --grid 40 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02n/bench_z40.json 2>/dev/null
python bench.py --workload elasticity --grid 56 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02n/bench_z56.json 2>/dev/null
python bench.py --grid 100 --steps 5 --warmup 2 > gpurun_out/r02n/bench_100.json 2>/dev/null
python bench.py --grid 192 --facto lu --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r02n/bench_192_lu.json 2>/dev/null
python bench.py --grid 100 --facto ldlt --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r02n/bench_100_ldlt.json 2>/dev/null
python bench.py --grid 100 --facto lu --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r02n/bench_100_lu.json 2>/dev/null
tail -3 gpurun_out/r02n/p200.log | cut -c1-600
for f in gpurun_out/r02n/bench_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d['config']['pct_of_mfma_f64_peak'], d['config']['residual'], d['roofline']['frac'])"; done
