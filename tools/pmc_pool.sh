#!/bin/bash
# PMC comparison of the update loop fed from cache (pool 4) and from HBM (pool 4096): read latency, translation
# misses, DRAM credit stalls.  Run on the GPU box.
export TMPDIR=/tmp
ROOT=$(pwd)
for set in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"; do
  for pool in 4 4096; do
    rm -rf /tmp/pp
    REPS=2 rocprofv3 --pmc $set --output-format csv -d /tmp/pp -- $ROOT/tools/bench_update 8192 16 128 $pool > /dev/null 2>&1
    echo "pool=$pool: $(python3 $ROOT/tools/pmc_sum.py /tmp/pp k_update)"
  done
done
