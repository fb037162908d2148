cd $GRAFT_REPO_ROOT
export OPENBLAS_NUM_THREADS=1 REF_ORDER_CONTIG=1
for a in "d 40 llt" "d 60 llt" "d 60 ldlt" "d 60 lu" "z 24 ldlt" "d 80 llt"; do
  set -- $a

  oracle/_ref/ref_harness_$1_ob_amd cmp rlap3d $2 $3 32 /dev/null 2>&1 | tail -3

done
