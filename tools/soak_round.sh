#!/bin/bash
# The round's soak of the run schedule (one kernel: every task of the run is a ticket of k_run_update): five configurations,
# >= 20 000 factorizations in all, each line of the logs a JSON summary of tools/soak_run.py (stops = factorizations in
# which a wait inside the run expired).  Usage (GPU box): bash tools/soak_round.sh <outdir>
out=${1:-gpurun_out/soak}
mkdir -p "$out"
cd "${GRAFT_REPO_ROOT:-.}"
run() { name=$1; shift; timeout 1500 python tools/soak_run.py "$@" > "$out/soak_$name.txt" 2>&1; tail -1 "$out/soak_$name.txt"; }
run d60_llt    -n 60  --facto 0 --reps 8000
run d100_llt   -n 100 --facto 0 --reps 3000
run d60_lu     -n 60  --facto 2 --reps 6000
run z32_ldlt   -n 32  --facto 1 --complex --reps 5000
run d60_ldlt   -n 60  --facto 1 --reps 3000
# round 6: the run is the default at 200^3 (its reader lists are built on the device): the metric's own configuration
run d200_llt   -n 200 --facto 0 --reps 40
# every factorization's factors hashed (bitwise determinism of the schedule, step by step), at sizes where the download is cheap
run det_d40_llt  -n 40 --facto 0 --reps 400 --check 1
run det_d40_ldlt -n 40 --facto 1 --reps 400 --check 1
run det_d48_lu   -n 48 --facto 2 --reps 200 --check 1
run det_z20_ldlt -n 20 --facto 1 --complex --reps 200 --check 1
