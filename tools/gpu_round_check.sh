mkdir -p gpurun_out/r02k
timeout 1200 python -m pytest tests -m gpu -q -k "ldlt or LDLT or sy or inertia or ref_caller or dist" > gpurun_out/r02k/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02k/pytest.log
tail -5 gpurun_out/r02k/pytest.log
for f in llt ldlt; do python bench.py --grid 100 --facto $f --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['metric'], d['value'], d['ms_per_step'], d['config']['residual'])"; done
PASTIX_AMD_DIAG_LDLT_OLD=1 python bench.py --grid 100 --facto ldlt --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('old diag kernel:', d['metric'], d['value'], d['ms_per_step'])"
