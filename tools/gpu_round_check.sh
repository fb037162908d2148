export TMPDIR=/tmp
bash tools/profile_round.sh 200 > gpurun_out/profile_200.log 2>&1
bash tools/profile_round.sh 100 > gpurun_out/profile_100.log 2>&1
NO_PMC=1 bash tools/profile_round.sh 48 --workload elasticity > gpurun_out/profile_48.log 2>&1
python bench.py --grid 100 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_100.json 2>/dev/null
python bench.py --grid 100 --facto ldlt --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_100_ldlt.json 2>/dev/null
python bench.py --grid 100 --facto lu --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_100_lu.json 2>/dev/null
python bench.py --grid 40 --workload elasticity --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_z40.json 2>/dev/null
python bench.py --grid 56 --workload elasticity --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_z56.json 2>/dev/null
python bench.py --grid 192 --facto lu --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_192_lu.json 2>/dev/null
for f in gpurun_out/bench_*.json gpurun_out/profile_200/bench_under_rocprof.json gpurun_out/profile_100/bench_under_rocprof.json gpurun_out/profile_48workloadelasticity/bench_under_rocprof.json; do python3 -c "
import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d.get('solve',{}).get('achieved'))"; done
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -1 | cut -c1-200
