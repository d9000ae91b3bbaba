#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh <tag>   -- kernel-trace stats + the two PMC traffic passes of the default bench,
# plus a kernel-trace of the HEADLINE region alone (no encoder-cached / whole-clip / aligner extras: only the warm-up and timed steps,
# all of the same shapes, so the per-kernel averages of its summary are directly the bench line's avg_launch_us)
TAG=$1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-bf16-run --no-raft-run --no-align-config3 > $OUT/bench_under_rocprof.json 2> $OUT/trace.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/headline -o ${TAG}h -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-cache-run --no-clip-run --no-align --no-bf16-run --no-raft-run > $OUT/bench_headline_under_rocprof.json 2> $OUT/headline.err &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o f -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 --align-iters 10 --no-cache-run --no-clip-run --no-bf16-run --no-raft-run --no-align-config3 > $OUT/fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o w -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 --align-iters 10 --no-cache-run --no-clip-run --no-bf16-run --no-raft-run --no-align-config3 > $OUT/write.log 2>&1
ls $OUT/*
