#!/bin/bash
# usage: tools/pmc_lab.sh <binary> <tag> <args...>  -- three PMC passes + a kernel trace of a lab binary (run on the GPU box)
BIN=$1; TAG=$2; shift 2
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $GRAFT_REPO_ROOT/$BIN "$@" > $OUT/trace.log 2>&1 &&
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p1 -o p -- $GRAFT_REPO_ROOT/$BIN "$@" > $OUT/p1.log 2>&1 &&
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/p2 -o p -- $GRAFT_REPO_ROOT/$BIN "$@" > $OUT/p2.log 2>&1 &&
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/p3 -o p -- $GRAFT_REPO_ROOT/$BIN "$@" > $OUT/p3.log 2>&1
find $OUT -name "*.csv" | head -30
