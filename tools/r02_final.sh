#!/bin/bash
# Round-2 final evidence (run from the repo root on the GPU box): bench lines + per-layer tables, rocprofv3 kernel stats,
# PMC traffic passes for the fp16 and the INT8 configuration.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
python3 $R/bench.py --steps 20 --warmup 5 --layers > $O/r02_bench_default.json 2> $O/r02_layers.txt; echo "bench default $?"
python3 $R/bench.py --int8 --steps 20 --warmup 5 --layers --no-cpu-baseline > $O/r02_int8_bench.json 2> $O/r02_int8_layers.txt; echo "bench int8 full $?"
python3 $R/bench.py --int8 --int8-recipe mixed --steps 20 --warmup 5 --layers --no-cpu-baseline > $O/r02_int8_mixed_bench.json 2> $O/r02_int8_mixed_layers.txt; echo "bench int8 mixed $?"
python3 $R/bench.py --int8 --int8-predequantize --steps 20 --warmup 5 --no-cpu-baseline > $O/r02_int8_predeq_bench.json 2> /dev/null; echo "bench int8 predeq $?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02f_kt -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r02f_kt.json 2> $O/r02f_kt.err; echo "kt $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02f_kt_i8 -o p -- python3 $R/bench.py --int8 --steps 20 --warmup 5 --no-cpu-baseline > $O/r02f_kt_i8.json 2> $O/r02f_kt_i8.err; echo "kt i8 $?"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/r02f_pmc_$ctr -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-int8-extra > $O/r02f_pmc_$ctr.log 2>&1; echo "pmc $ctr $?"
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/r02f_pmc_i8_$ctr -o p -- python3 $R/bench.py --int8 --steps 3 --warmup 1 --no-cpu-baseline > $O/r02f_pmc_i8_$ctr.log 2>&1; echo "pmc i8 $ctr $?"
done
cd $R
python3 $R/bench.py --steps 20 --warmup 5 --cpu-protocol full --no-int8-extra > $O/r02_bench_cpu_full_protocol.json 2> /dev/null; echo "bench cpu-full $?"
