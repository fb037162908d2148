#!/bin/bash
# After `gpurun -- 'bash tools/round_final.sh'`: fold gpurun_out/ into profiles/rNN/ (usage: collect_profiles.sh r04)
R=${1:?round directory name, e.g. r04}
cd "$(dirname "$0")/.."
python3 tools/make_profile_json.py $R 200 100 48workloadelasticity > /dev/null
P=profiles/$R
for t in 200 100; do
  cp gpurun_out/profile_$t/kernel_stats.csv $P/kernel_stats_bench_${t}cube.csv
  cp gpurun_out/profile_$t/bench_under_rocprof.json $P/bench_${t}cube_under_rocprof.json
done
cp gpurun_out/profile_48workloadelasticity/kernel_stats.csv $P/kernel_stats_bench_z_elasticity_48.csv
cp gpurun_out/profile_48workloadelasticity/bench_under_rocprof.json $P/bench_z_elasticity_48_under_rocprof.json
cp gpurun_out/bench_default.json $P/bench_default_run.json
cp gpurun_out/bench_f32_200.json $P/bench_f32_200cube.json
cp gpurun_out/bench_f32_100.json $P/bench_f32_100cube.json
cp gpurun_out/bench_100_ldlt.json gpurun_out/bench_100_lu.json gpurun_out/bench_z40.json gpurun_out/bench_z56.json $P/
[ -d gpurun_out/final ] && cp gpurun_out/final/*.txt gpurun_out/final/*.json $P/ 2>/dev/null
cat gpurun_out/profile_200/source_sha.txt
