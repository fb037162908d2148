#!/bin/bash
# Run ON THE GPU BOX in two calls (gpurun -- "bash tools/round_final.sh a", then "... b"): the round's final measurements.
#   a: the three profile passes, the default bench run, the side configurations
#   b: the size sweep, run against level schedule, the per-level / per-ticket timelines of the mid sizes, the
#      reference-as-caller timings, the analysis phases and the loopback run
# Every step has its own time limit.  tools/collect_profiles.sh rNN folds gpurun_out/ into profiles/rNN/.
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
J='import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%-60s %9.1f GFLOP/s %9.2f ms" % (d["metric"][:60], d["value"], d["ms_per_step"]))'
if [ "$1" = "a" ]; then
timeout 900 bash tools/profile_round.sh 200 > gpurun_out/profile_200.log 2>&1
timeout 600 bash tools/profile_round.sh 100 > gpurun_out/profile_100.log 2>&1
timeout 600 bash tools/profile_round.sh 48 --workload elasticity > gpurun_out/profile_z48.log 2>&1
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
timeout 300 python bench.py --dtype f32 --no-cpu-baseline --no-other-configs --steps 2 > gpurun_out/bench_f32_200.json 2>/dev/null
timeout 200 python bench.py --dtype f32 --grid 100 --no-cpu-baseline --no-other-configs --steps 5 > gpurun_out/bench_f32_100.json 2>/dev/null
timeout 200 python bench.py --grid 100 --facto ldlt --no-cpu-baseline --no-other-configs --steps 5 > gpurun_out/bench_100_ldlt.json 2>/dev/null
timeout 200 python bench.py --grid 100 --facto lu --no-cpu-baseline --no-other-configs --steps 5 > gpurun_out/bench_100_lu.json 2>/dev/null
timeout 200 python bench.py --grid 40 --workload elasticity --no-cpu-baseline --no-other-configs --steps 5 > gpurun_out/bench_z40.json 2>/dev/null
timeout 200 python bench.py --grid 56 --workload elasticity --no-cpu-baseline --no-other-configs --steps 3 > gpurun_out/bench_z56.json 2>/dev/null
tail -c 600 gpurun_out/bench_default.json
exit 0
fi
# ---- b ----
timeout 900 bash tools/sweep_sizes.sh > $O/sweep_sizes.txt 2>&1
( echo "# level schedule (PASTIX_AMD_RUN=0) against the run schedule on ONE plan, same box: tools/dev_run_ab.py"; timeout 600 python tools/dev_run_ab.py -n 40 60 80 100 130 160 --reps 3 --nocheck 2>&1 | grep -E "run=|speedup" ) > $O/run_vs_level.txt
( for f in ldlt lu; do for g in 60 100 130; do for r in 0 1; do echo -n "run=$r "; PASTIX_AMD_RUN=$r timeout 200 python bench.py --grid $g --facto $f --steps 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "$J"; done; done; done
  for g in 32 48; do for r in 0 1; do echo -n "run=$r "; PASTIX_AMD_RUN=$r timeout 200 python bench.py --grid $g --workload elasticity --steps 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "$J"; done; done ) >> $O/run_vs_level.txt 2>&1
for n in 60 100; do
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr$n && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$n -- python3 $GRAFT_REPO_ROOT/tools/dev_bench.py -n $n --reps 2 > /dev/null 2>&1 )
  python tools/level_rows.py /tmp/tr$n 30 > $O/level_timeline_d$n.txt 2>&1
  PASTIX_AMD_DEV=run_prof=/tmp/prof_$n.bin timeout 300 python tools/dev_run_ab.py -n $n --reps 2 --nocheck > /dev/null 2>&1
  python tools/run_prof.py /tmp/prof_$n.bin 40 >> $O/level_timeline_d$n.txt 2>&1
done
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/trz && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trz -- python3 $GRAFT_REPO_ROOT/bench.py --grid 48 --workload elasticity --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null 2>&1 )
python tools/level_rows.py /tmp/trz 30 > $O/level_timeline_z48.txt 2>&1
PASTIX_AMD_DEV=run_prof=/tmp/prof_z.bin timeout 300 python bench.py --grid 48 --workload elasticity --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
python tools/run_prof.py /tmp/prof_z.bin 40 >> $O/level_timeline_z48.txt 2>&1
[ -x tools/bench_diag ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDIAG_PROFILE -I pastix_amd/csrc -I include -o tools/bench_diag tools/bench_diag.hip
( for w in 24 64 128; do timeout 60 ./tools/bench_diag $w 1 | tail -1; done; timeout 60 ./tools/bench_diag 128 2048 | tail -1 ) > $O/bench_diag.txt 2>&1
if [ -x oracle/_ref/ref_harness_d_ob_amd ]; then
  ( export OPENBLAS_NUM_THREADS=1
    echo "## part 1: separator nodes numbered lexicographically"
    for n in 60 80 100; do echo "== ref caller rlap3d $n"; timeout 300 oracle/_ref/ref_harness_d_ob_amd amd rlap3d $n llt 1 /dev/null 2>/dev/null | tail -1; done
    echo "## part 2 (REF_ORDER_CONTIG=1): separator nodes numbered along their low-side neighbours"
    for n in 60 80 100; do echo "== ref caller rlap3d $n contiguous separators"; REF_ORDER_CONTIG=1 timeout 300 oracle/_ref/ref_harness_d_ob_amd amd rlap3d $n llt 1 /dev/null 2>/dev/null | tail -1; done
    echo "## the engine on its own layout of the same matrices (tools/dev_bench.py)"
    for n in 60 80 100; do timeout 200 python tools/dev_bench.py -n $n --reps 3 2>/dev/null | tail -1; done
    echo "## cmp mode: factors of the engine against the reference's CPU engine, entry-wise"
    for c in "d 60 llt" "d 80 llt" "d 100 llt" "d 100 lu" "z 32 ldlt"; do set -- $c
      echo "== cmp $c"; REF_ORDER_CONTIG=1 timeout 600 oracle/_ref/ref_harness_$1_ob_amd cmp rlap3d $2 $3 32 /dev/null 64 128 2>/dev/null | grep '"cmp"' | tail -1; done ) > $O/refcaller_timing.txt 2>&1
fi
timeout 300 python tools/one_shot_timing.py 100 > $O/one_shot_100cube.json 2> /dev/null
if [ -x oracle/_ref/ref_harness_d_ob_amd ]; then      # where the time of a fragmented layout goes: the run's tickets by class
  OPENBLAS_NUM_THREADS=1 PASTIX_AMD_DEV=run_prof=/tmp/prof_lex.bin timeout 300 oracle/_ref/ref_harness_d_ob_amd amd rlap3d 100 llt 1 /dev/null 64 128 > /dev/null 2>&1
  python tools/run_prof.py /tmp/prof_lex.bin 10 > $O/run_prof_refcaller_lex100.txt 2>&1
fi
PASTIX_AMD_DEV=plan_timing timeout 300 python tools/plan_timing.py 200 > $O/analysis_timing_200cube.txt 2>&1
timeout 900 python tools/loopback_scale.py 200 4 2 > $O/loopback_200cube_4ranks.json 2> $O/loopback_200.err
tail -3 $O/sweep_sizes.txt
