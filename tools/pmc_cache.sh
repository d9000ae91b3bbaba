#!/bin/bash
# usage (on the GPU box): tools/pmc_cache.sh <tag> <python script> [args...]  -- L2 / L1 / fabric counters of a python tool
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
S=$GRAFT_REPO_ROOT/$1; shift
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --kernel-trace --output-format csv -d $OUT/p1 -o p -- python3 $S "$@" > $OUT/p1.log 2>&1 &&
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUSY_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p2 -o p -- python3 $S "$@" > $OUT/p2.log 2>&1 &&
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_sum --kernel-trace --output-format csv -d $OUT/p3 -o p -- python3 $S "$@" > $OUT/p3.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE TCC_TAG_STALL_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum --kernel-trace --output-format csv -d $OUT/p4 -o p -- python3 $S "$@" > $OUT/p4.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT
