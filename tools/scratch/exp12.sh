#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for cfg in "-n 40 --facto 2" "-n 48 --facto 2" "-n 20 --facto 1 --complex" "-n 40 --facto 0"; do
  for k in 1 0; do
    echo "== $cfg onek=$k"; PASTIX_AMD_DEV="onek=$k" timeout 600 python tools/soak_run.py $cfg --reps 120 --check 1 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('stops %d errors %d distinct digests %d (of %d checks) median %.2f ms' % (d['stops'], d['errors'], d['distinct_digests'], d['reps'], d['median_ms']))"
  done
done
