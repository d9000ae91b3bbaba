#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh <tag>   -- kernel-trace stats + the two PMC traffic passes of the default bench
TAG=$1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o f -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 --align-iters 10 --no-cache-run > $OUT/fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o w -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 --align-iters 10 --no-cache-run > $OUT/write.log 2>&1
ls $OUT/*
