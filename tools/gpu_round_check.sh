export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -q -x 2>&1 | tail -1
rm -rf /tmp/le
rocprofv3 --kernel-trace --output-format csv -d /tmp/le -- python3 tools/dev_bench.py -n 100 --reps 2 > /tmp/le.log 2>&1
python3 tools/leaf_timeline.py /tmp/le 40 | grep -E "diag|busy"
for n in 60 100; do python tools/dev_bench.py -n $n --reps 4 2>&1 | tail -1 | cut -c1-75; done
python bench.py --grid 100 --facto ldlt --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['metric'], d['value'], d['ms_per_step'])"
