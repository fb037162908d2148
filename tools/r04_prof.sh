cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
export PASTIX_AMD_RUN_TIMEOUT=5
timeout 120 python tools/dev_run_ab.py -n 12 20 30 --reps 1 2>&1 | grep -E "bitwise|failed|rror"
for n in 60 100; do
  PASTIX_AMD_RUN_PROF=/tmp/prof_$n.bin timeout 300 python tools/dev_run_ab.py -n $n --nocheck --reps 2 2>&1 | grep -v amdgpu.ids
  python tools/run_prof.py /tmp/prof_$n.bin > gpurun_out/r04/run_prof_$n.txt 2>&1
  head -12 gpurun_out/r04/run_prof_$n.txt
done
for cfg in "120 40" "60 20" "200 60" "0 0"; do
  set -- $cfg
  echo "== delays D=$1 T=$2"; PASTIX_AMD_RUN_DELAY_D=$1 PASTIX_AMD_RUN_DELAY_T=$2 timeout 300 python tools/dev_run_ab.py -n 60 100 130 --nocheck --reps 3 2>&1 | grep -E "speedup|run=1"
done
echo "== dw 4"; timeout 300 python tools/dev_run_ab.py -n 60 100 130 --nocheck --reps 3 --dw 4 2>&1 | grep -E "speedup|run=1"
echo "== maxc 8"; timeout 300 python tools/dev_run_ab.py -n 60 100 130 --nocheck --reps 3 --maxc 8 2>&1 | grep -E "speedup|run=1"
