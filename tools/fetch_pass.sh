#!/bin/bash
# usage (on the GPU box): tools/fetch_pass.sh <tag> <python script> [args...]  -- one FETCH_SIZE PMC pass of a python tool, per-kernel means
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/fetch_$TAG
mkdir -p $OUT
S=$GRAFT_REPO_ROOT/$1; shift
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/p4 -o p -- python3 $S "$@" > $OUT/p.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT
