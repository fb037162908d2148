#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp2
for k in 0 1; do
  PASTIX_AMD_RUN_ROOM=0 PASTIX_AMD_RUN_ONEK=$k timeout 900 python tools/soak_run.py -n 100 --reps 1500 --tag room0_onek$k > gpurun_out/exp2/soak100_room0_onek$k.log 2>&1
  tail -4 gpurun_out/exp2/soak100_room0_onek$k.log
done
PASTIX_AMD_RUN_ONEK=1 timeout 900 python tools/soak_run.py -n 60 --reps 20000 --tag onek1 > gpurun_out/exp2/soak60_onek1.log 2>&1
tail -3 gpurun_out/exp2/soak60_onek1.log
