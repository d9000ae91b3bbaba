#!/bin/bash
# usage (on the GPU box): tools/pmc_py.sh <tag> <python script> [args...]  -- kernel trace + PMC passes of a python tool
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
S=$GRAFT_REPO_ROOT/$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $S "$@" > $OUT/trace.log 2>&1 &&
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/p1 -o p -- python3 $S "$@" > $OUT/p1.log 2>&1 &&
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/p2 -o p -- python3 $S "$@" > $OUT/p2.log 2>&1 &&
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/p3 -o p -- python3 $S "$@" > $OUT/p3.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/p4 -o p -- python3 $S "$@" > $OUT/p4.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT
