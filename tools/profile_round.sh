#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/profile_round.sh 200'): everything the round's profile of one workload
# consists of, every counter set in its own rocprofv3 pass with no tracing beside it:
#   1. kernel-trace statistics of one bench.py step                       -> kernel_stats.csv, bench_under_rocprof.json
#   2. HBM traffic of the bulk update kernels (FETCH_SIZE, WRITE_SIZE)    -> sum_FETCH_SIZE.json, sum_WRITE_SIZE.json
#   3. MFMA-pipe / CU busy cycles of the same kernels                    -> sum_busy.json
#   4. L2 hit rate                                                       -> sum_l2.json
# usage: profile_round.sh GRID [extra bench.py args, e.g. --workload elasticity]     (NO_PMC=1: step 1 only)
# Outputs under gpurun_out/profile_<tag>/ ; tools/make_profile_json.py folds them into profiles/rNN/.
G=${1:-200}; shift
EXTRA="$@"
TAG=$G$(echo "$EXTRA" | tr -d ' -')
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profile_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# the bulk update kernels: the run launch (k_run_update: the thin levels in one dependency-driven launch, where the run
# schedule is built) and the per-level launches below it, k_update<0> and -- where a slot has quadrant tasks --
# k_update_small<0> right behind it
KERNELS="k_run_update,k_update<0>,k_update_small<0>"
ARGS="bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs --grid $G $EXTRA"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats
if [ -z "$NO_PMC" ]; then
# The counter passes run the DEFAULT schedule for real LLt / LDLt: since round 5 every task of their run is a ticket of ONE
# kernel (k_run_update), so the launch needs nothing beside it on the chip and rocprofv3 --pmc, which serializes kernel
# launches, can count it.  LU and complex plans still have the resident diagonal kernel beside the tickets' launch: under the
# counters they take the level-by-level schedule (same kernels' bodies, same tasks, same flops).
case "$EXTRA" in *elasticity*|*lu*) export PASTIX_AMD_RUN=0;; esac
pass() {   # name, counters
  rm -rf /tmp/pmc_pass
  timeout 600 rocprofv3 --pmc $2 --output-format csv -d /tmp/pmc_pass -- python3 $ARGS > $OUT/bench_pmc_$1.json 2> $OUT/pmc_$1.err
  python3 tools/pmc_sum.py /tmp/pmc_pass "$KERNELS" > $OUT/sum_$1.json
  rm -rf /tmp/pmc_pass
}
pass FETCH_SIZE "FETCH_SIZE"
pass WRITE_SIZE "WRITE_SIZE"
pass busy "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64"
pass waves "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
pass l2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
unset PASTIX_AMD_RUN
fi
python3 -c "import bench; print(bench.engine_source_sha())" > $OUT/source_sha.txt
cat $OUT/sum_*.json 2>/dev/null
head -8 $OUT/kernel_stats.csv
tail -1 $OUT/bench_under_rocprof.json
