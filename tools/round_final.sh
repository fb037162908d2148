#!/bin/bash
# Run ON THE GPU BOX (gpurun -- "bash tools/round_final.sh"): the round's final measurements -- the three profile passes, the
# default bench run and the side configurations -- into gpurun_out/; tools/make_profile_json.py + copies fill profiles/rNN/
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh 200 > gpurun_out/profile_200.log 2>&1
bash tools/profile_round.sh 100 > gpurun_out/profile_100.log 2>&1
bash tools/profile_round.sh 48 --workload elasticity > gpurun_out/profile_z48.log 2>&1
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python bench.py --dtype f32 --no-cpu-baseline --no-other-configs --steps 2 > gpurun_out/bench_f32_200.json 2>/dev/null
python bench.py --dtype f32 --grid 100 --no-cpu-baseline --no-other-configs --steps 5 > gpurun_out/bench_f32_100.json 2>/dev/null
python bench.py --grid 100 --facto ldlt --no-cpu-baseline --no-other-configs --steps 5 > gpurun_out/bench_100_ldlt.json 2>/dev/null
python bench.py --grid 100 --facto lu --no-cpu-baseline --no-other-configs --steps 5 > gpurun_out/bench_100_lu.json 2>/dev/null
python bench.py --grid 40 --workload elasticity --no-cpu-baseline --no-other-configs --steps 5 > gpurun_out/bench_z40.json 2>/dev/null
python bench.py --grid 56 --workload elasticity --no-cpu-baseline --no-other-configs --steps 3 > gpurun_out/bench_z56.json 2>/dev/null
tail -c 600 gpurun_out/bench_default.json
