#!/bin/bash
# Run ON THE GPU BOX: bench.py over a range of sizes / factorizations (one line each: value, time, bulk-kernel fraction,
# residual, device solve time) -- reference data beside the configurations of BASELINE.json
cd $GRAFT_REPO_ROOT
one() { python bench.py "$@" --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-92s %9.1f GFLOP/s %9.2f ms  frac %.3f  resid %.1e  solve %.2f ms' % (d['metric'][:92], d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['residual'], 1e3*d['solve']['device_s']))"; }
for g in 40 60 80 130 160; do one --grid $g --steps 5; done
for f in ldlt lu; do for g in 60 160; do one --grid $g --facto $f --steps 3; done; done
for g in 24 32 64; do one --grid $g --workload elasticity --steps 5; done
for g in 60 160; do one --grid $g --dtype f32 --steps 3; done
