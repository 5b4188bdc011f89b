#!/bin/bash
# usage: [BENCH_ARGS=--int8] tools/pmc_pass.sh <tag> <counter list...>   (run from the repo root on the GPU box)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
echo "pmc $tag exit $?"
