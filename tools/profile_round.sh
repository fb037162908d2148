#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/profile_round.sh 200'): kernel-trace statistics and the two HBM traffic
# passes (FETCH_SIZE, WRITE_SIZE; separate --pmc runs, no other tracing) of one bench.py step.
# usage: profile_round.sh GRID [extra bench.py args, e.g. --workload elasticity]
# Outputs under gpurun_out/profile_<tag>/ ; tools/make_traffic_json.py folds them into profiles/rNN/.
G=${1:-200}; shift
EXTRA="$@"
TAG=$G$(echo "$EXTRA" | tr -d ' -')
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profile_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 1 --warmup 0 --no-cpu-baseline --grid $G $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
if [ -z "$NO_PMC" ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $ARGS > $OUT/bench_pmc_$c.json 2> $OUT/pmc_$c.err
  python3 tools/pmc_sum.py $OUT/pmc_$c "k_update<8, 0>" > $OUT/sum_$c.json
  rm -rf $OUT/pmc_$c
done
fi
rm -rf $OUT/stats
python3 -c "import bench; print(bench.engine_source_sha())" > $OUT/source_sha.txt
cat $OUT/sum_*.json 2>/dev/null
head -8 $OUT/kernel_stats.csv
tail -1 $OUT/bench_under_rocprof.json
