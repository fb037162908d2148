export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_solve_driver.py tests/test_gpu_dist.py tests/test_gpu_ref_caller.py -q -x -k "z or Z or complex or config5 or young or herm or ldlh" 2>&1 | tail -1 | cut -c1-200
for rep in 1 2; do for w in 40 48 56; do
echo -n "new  z$w: "; python bench.py --grid $w --workload elasticity --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
echo -n "prev z$w: "; PASTIX_AMD_LIB=$PWD/tools/libpastix_amd_prev.so python bench.py --grid $w --workload elasticity --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
