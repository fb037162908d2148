#!/bin/bash
# Run ON THE GPU BOX: kernel statistics and the matrix-pipe counters of the fp32 engine at 200^3 (bench.py --dtype f32)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/profile_f32_200
mkdir -p $O
ARGS="bench.py --dtype f32 --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs"
rm -rf /tmp/pmc_pass
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmc_pass -- python3 $ARGS > $O/bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/pmc_pass -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf /tmp/pmc_pass
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_pass -- python3 $ARGS > $O/bench_pmc.json 2>/dev/null
python3 tools/pmc_sum.py /tmp/pmc_pass "k_update_s<0>" > $O/sum_busy.json
rm -rf /tmp/pmc_pass
cat $O/sum_busy.json; head -5 $O/kernel_stats.csv | cut -c1-150
