cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
export PASTIX_AMD_RUN_TIMEOUT=5
for ps in 0 512 496; do
echo "== persist $ps"; PASTIX_AMD_RUN_PERSIST=$ps timeout 300 python tools/dev_run_ab.py -n 20 60 100 130 --reps 3 2>&1 | grep -E "speedup|run=1"
done
PASTIX_AMD_RUN_PERSIST=512 PASTIX_AMD_RUN_PROF=/tmp/prof_100.bin timeout 300 python tools/dev_run_ab.py -n 100 --reps 2 --nocheck 2>&1 | grep -v amdgpu.ids
python tools/run_prof.py /tmp/prof_100.bin > gpurun_out/r04/run_prof_100p.txt 2>&1
head -14 gpurun_out/r04/run_prof_100p.txt
