export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_solve_driver.py tests/test_gpu_ref_caller.py tests/test_gpu_dist.py -q -x 2>&1 | tail -1 | cut -c1-200
for rep in 1 2; do
echo -n "new  ldlt 100: "; python bench.py --grid 100 --facto ldlt --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
echo -n "prev ldlt 100: "; PASTIX_AMD_LIB=$PWD/tools/libpastix_amd_prev.so python bench.py --grid 100 --facto ldlt --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
echo -n "new  ldlt 60: "; python bench.py --grid 60 --facto ldlt --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
echo -n "prev ldlt 60: "; PASTIX_AMD_LIB=$PWD/tools/libpastix_amd_prev.so python bench.py --grid 60 --facto ldlt --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
