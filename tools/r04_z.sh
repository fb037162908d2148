cd $GRAFT_REPO_ROOT
one() { python bench.py "$@" --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-70s %9.1f GFLOP/s %9.2f ms  kernel %s frac %.3f  resid %.1e' % (d['metric'][:70], d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['config']['residual']))"; }
for c in 512 1024 2048 4096; do for g in 32 48; do echo -n "chunk=$c "; one --grid $g --workload elasticity --steps 3 --chunk $c; done; done
PASTIX_AMD_RUN_PROF=/tmp/prof_z.bin python bench.py --grid 48 --workload elasticity --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
python tools/run_prof.py /tmp/prof_z.bin 2>&1 | head -14
