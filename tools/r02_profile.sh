#!/bin/bash
# Round-2 profile collection (run from the repo root on the GPU box): kernel-trace stats of the default bench command,
# then FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs) for both HG tile orders.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_kt -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r02_kt.json 2> $R/gpurun_out/r02_kt.err
echo "kt exit $?"
for nt in 0 1; do
  export HDRTV_VARIANTS=pglds_nt_slow=$nt
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/gpurun_out/r02_pmc_${ctr}_nt$nt -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-int8-extra > $R/gpurun_out/r02_pmc_${ctr}_nt$nt.log 2>&1
    echo "pmc $ctr nt=$nt exit $?"
  done
done
unset HDRTV_VARIANTS
