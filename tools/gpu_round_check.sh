export TMPDIR=/tmp
for rep in 1 2; do
for n in 60 100 160; do
echo -n "prio   N=$n: "; python tools/dev_bench.py -n $n --reps 5 2>&1 | tail -1 | cut -c1-60
echo -n "noprio N=$n: "; PASTIX_AMD_LIB=$PWD/tools/libpastix_amd_noprio.so python tools/dev_bench.py -n $n --reps 5 2>&1 | tail -1 | cut -c1-60
done; done
for w in 40 48; do
echo -n "prio   z$w: "; python bench.py --grid $w --workload elasticity --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
echo -n "noprio z$w: "; PASTIX_AMD_LIB=$PWD/tools/libpastix_amd_noprio.so python bench.py --grid $w --workload elasticity --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
