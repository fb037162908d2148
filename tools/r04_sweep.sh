cd $GRAFT_REPO_ROOT
export PASTIX_AMD_RUN_TIMEOUT=10
for cfg in "--look 512" "--look 1024" "--look 2048" "--look 4096"; do
  echo "== $cfg"; timeout 600 python tools/dev_run_ab.py -n 60 80 100 130 160 --nocheck --reps 3 $cfg 2>&1 | grep -E "run="
done
