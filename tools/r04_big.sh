cd $GRAFT_REPO_ROOT
export PASTIX_AMD_RUN_TIMEOUT=20
PASTIX_AMD_PLAN_TIMING=1 timeout 900 python tools/dev_run_ab.py -n 160 --reps 2 --nocheck 2>&1 | grep -v amdgpu.ids | tail -30
PASTIX_AMD_PLAN_TIMING=1 timeout 1500 python tools/dev_run_ab.py -n 200 --reps 2 --nocheck 2>&1 | grep -v amdgpu.ids | tail -30
