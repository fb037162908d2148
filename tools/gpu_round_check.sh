export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -q -x 2>&1 | tail -2
rm -rf /tmp/zp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/zp -- python3 bench.py --grid 100 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/zp.json 2>/dev/null
python3 -c "
import json; d=json.loads(open('/tmp/zp.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
head -6 $(find /tmp/zp -name "*kernel_stats.csv" | head -1) | cut -c1-60,100-175
for n in 60 100 160; do python tools/dev_bench.py -n $n --reps 4 2>&1 | tail -1; done
python bench.py --grid 100 --facto ldlt --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['metric'], d['value'], d['ms_per_step'])"
