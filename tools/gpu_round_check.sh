export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -q -x 2>&1 | tail -1 | cut -c1-200
for rep in 1 2; do
for n in 60 100 160; do
echo -n "new  N=$n: "; python tools/dev_bench.py -n $n --reps 5 2>&1 | tail -1 | cut -c1-60
echo -n "prev N=$n: "; PASTIX_AMD_LIB=$PWD/tools/libpastix_amd_prev.so python tools/dev_bench.py -n $n --reps 5 2>&1 | tail -1 | cut -c1-60
done; done
for f in ldlt lu; do
echo -n "new  $f: "; python bench.py --grid 100 --facto $f --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
echo -n "prev $f: "; PASTIX_AMD_LIB=$PWD/tools/libpastix_amd_prev.so python bench.py --grid 100 --facto $f --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
