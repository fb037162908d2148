#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for lib in alt_nopark alt_nops; do
  echo "== $lib LU 48"; PASTIX_AMD_LIB=$PWD/pastix_amd/lib/$lib.so timeout 600 python tools/soak_run.py -n 48 --facto 2 --reps 120 --check 1 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('stops %d errors %d distinct digests %d (of %d checks) median %.2f ms' % (d['stops'], d['errors'], d['distinct_digests'], d['reps'], d['median_ms']))"
done
