#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export OPENBLAS_NUM_THREADS=1 REF_ORDER_CONTIG=1
for k in 0 1; do
  for rep in 1 2 3 4 5 6 7 8 9 10 11 12; do
    r=$(PASTIX_AMD_DEV="gather=-1,onek=$k" timeout 300 oracle/_ref/ref_harness_d_ob_amd cmp rlap3d 60 lu 32 /dev/null 2>/dev/null | grep '"cmp"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2e %d' % (max(d['rel_L'],d['rel_U']), d['worst_cblk']))")
    echo "onek=$k rep $rep: $r"
  done
done
