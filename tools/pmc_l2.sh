#!/bin/bash
# L2 hit rate of the bulk update kernel in one factorization (run on the GPU box): bash tools/pmc_l2.sh [grid]
G=${1:-160}
export TMPDIR=/tmp
ROOT=$(pwd)
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  rm -rf /tmp/pl2
  rocprofv3 --pmc $set --output-format csv -d /tmp/pl2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --grid $G > /dev/null 2>&1
  python3 $ROOT/tools/pmc_sum.py /tmp/pl2 "k_update<8, 0>"
done
