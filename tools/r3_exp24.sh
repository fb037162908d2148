#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf /tmp/pmc_pass
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_pass -- python3 bench.py --dtype f32 --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > gpurun_out/f32_pmc_bench.json 2>/dev/null
python3 tools/pmc_sum.py /tmp/pmc_pass "k_update_s<0>" > gpurun_out/f32_pmc_sum.json
cat gpurun_out/f32_pmc_sum.json
rm -rf /tmp/pmc_pass
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmc_pass -- python3 bench.py --dtype f32 --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > gpurun_out/f32_stats_bench.json 2>/dev/null
cp $(find /tmp/pmc_pass -name "*kernel_stats.csv" | head -1) gpurun_out/f32_kernel_stats_200.csv
head -6 gpurun_out/f32_kernel_stats_200.csv | cut -c1-160
