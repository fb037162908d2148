cd $GRAFT_REPO_ROOT
one() { python bench.py "$@" --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-70s %9.1f GFLOP/s %9.2f ms  kernel %s frac %.3f  resid %.1e' % (d['metric'][:70], d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['config']['residual']))"; }
for f in $1; do for g in $2; do for r in 0 1; do echo -n "run=$r "; PASTIX_AMD_RUN=$r one --grid $g --facto $f --steps 3; done; done; done
