#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export OPENBLAS_NUM_THREADS=1 REF_ORDER_CONTIG=1
for g in 0 -1; do
  bad=0
  for rep in 1 2 3 4 5 6 7 8; do
    r=$(PASTIX_AMD_DEV="gather=$g" timeout 300 oracle/_ref/ref_harness_d_ob_amd cmp rlap3d 60 lu 32 /dev/null 2>/dev/null | grep '"cmp"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2e %d' % (max(d['rel_L'],d['rel_U']), d['worst_cblk']))")
    echo "gather=$g rep $rep: $r"
  done
done
unset REF_ORDER_CONTIG
for g in 0 -1; do
  for rep in 1 2 3 4; do
    r=$(PASTIX_AMD_DEV="gather=$g" timeout 300 oracle/_ref/ref_harness_d_ob_amd cmp rlap3d 60 lu 32 /dev/null 2>/dev/null | grep '"cmp"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2e %d' % (max(d['rel_L'],d['rel_U']), d['worst_cblk']))")
    echo "lexicographic gather=$g rep $rep: $r"
  done
done
