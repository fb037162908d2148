#!/bin/bash
# Occupancy / MFMA-pipe utilisation of the bulk update kernel over one factorization (run on the GPU box).
G=${1:-160}
export TMPDIR=/tmp
ROOT=$(pwd)
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  rm -rf /tmp/pl2
  rocprofv3 --pmc $set --output-format csv -d /tmp/pl2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --grid $G > /dev/null 2>&1
  python3 $ROOT/tools/pmc_sum.py /tmp/pl2 "k_update<8, 0>"
done
