#!/bin/bash
# Round-5 evidence (run from the repo root on the GPU box): bench lines + per-layer tables, rocprofv3 kernel stats, PMC traffic
# passes (fp16 and INT8), MFMA-busy / effective clock and the SQ wave-cycle breakdown of every kernel.
# usage: tools/r05_final.sh bench   (the bench lines and per-layer tables)   |   tools/r05_final.sh prof   (rocprofv3 passes + summaries)
# (two gpurun calls: together they exceed one call's 20-minute limit)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
B="--steps 20 --warmup 5"
if [ "${1:-bench}" = bench ]; then
python3 $R/bench.py $B --layers > $O/r05_bench_default.json 2> $O/r05_layers.txt; echo "bench default $?"
python3 $R/bench.py $B --lanes 2 --no-cpu-baseline --no-dispatcher --no-int8-extra > $O/r05_bench_two_lanes.json 2> /dev/null; echo "bench two lanes $?"
HDRTV_VARIANTS=le_rows=0 python3 $R/bench.py $B --layers --no-cpu-baseline --no-dispatcher --no-int8-extra > $O/r05_bench_le_rows_off.json 2> $O/r05_layers_le_rows_off.txt; echo "bench le_rows=0 $?"
python3 $R/bench.py --int8 $B --layers --no-cpu-baseline --no-dispatcher > $O/r05_int8_bench.json 2> $O/r05_int8_layers.txt; echo "bench int8 full $?"
python3 $R/bench.py --int8 --int8-recipe mixed $B --no-cpu-baseline --no-dispatcher > $O/r05_int8_mixed_bench.json 2> /dev/null; echo "bench int8 mixed $?"
python3 $R/bench.py --int8 --int8-predequantize $B --no-cpu-baseline --no-dispatcher > $O/r05_int8_predeq_bench.json 2> /dev/null; echo "bench int8 predeq $?"
HDRTV_VARIANTS=le_rows_i8=0 python3 $R/bench.py --int8 $B --layers --no-cpu-baseline --no-dispatcher > $O/r05_int8_rows_i8_off_bench.json 2> $O/r05_int8_rows_i8_off_layers.txt; echo "bench int8, row kernels in the fake-quant form $?"
HDRTV_VARIANTS=le_rows_i8=0,le_rows_fq=0 python3 $R/bench.py --int8 $B --no-cpu-baseline --no-dispatcher > $O/r05_int8_per_layer_bench.json 2> /dev/null; echo "bench int8 per layer $?"
python3 $R/bench.py --height 1080 --width 1920 $B --no-cpu-baseline --no-dispatcher --no-int8-extra > $O/r05_bench_1080p.json 2> /dev/null; echo "bench 1080p $?"
python3 $R/bench.py --steps 1500 --warmup 5 --no-cpu-baseline --no-dispatcher --no-int8-extra > $O/r05_soak_1500.json 2> /dev/null; echo "soak $?"
python3 $R/tools/fp32_layers.py --variant f32_mfma=0,1 --top 40 > $O/r05_fp32_layers.txt 2>&1; echo "fp32 layers $?"
python3 $R/tools/fp32_layers.py --size 2160x3840 --variant f32_mfma=1 --top 12 > $O/r05_fp32_layers_4k.txt 2>&1; echo "fp32 layers 4K $?"
exit 0
fi
cd /tmp && export TMPDIR=/tmp
# every pass with one frame in flight (the default) and without the `two_lanes` leg: with two lanes a kernel's begin-to-end time
# includes the share of the device the other lane's kernel holds.  The two-lane mode is traced once: r05_kernel_stats_2lanes.csv.
Q="--no-cpu-baseline --no-int8-extra --no-dispatcher --no-latency-tail --no-two-lanes"
Q2="$Q --lanes 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05f_kt -o p -- python3 $R/bench.py $B $Q > $O/r05f_kt.json 2> $O/r05f_kt.err; echo "kt $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05f_kt2 -o p -- python3 $R/bench.py $B $Q2 > $O/r05f_kt2.json 2> $O/r05f_kt2.err; echo "kt, two lanes $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05f_kt_i8 -o p -- python3 $R/bench.py --int8 $B $Q > $O/r05f_kt_i8.json 2> $O/r05f_kt_i8.err; echo "kt i8 $?"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/r05f_pmc_$ctr -o p -- python3 $R/bench.py --steps 3 --warmup 1 $Q > $O/r05f_pmc_$ctr.log 2>&1; echo "pmc $ctr $?"
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/r05f_pmc_i8_$ctr -o p -- python3 $R/bench.py --int8 --steps 3 --warmup 1 $Q > $O/r05f_pmc_i8_$ctr.log 2>&1; echo "pmc i8 $ctr $?"
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/r05f_mfma -o p -- python3 $R/bench.py --steps 3 --warmup 1 $Q > $O/r05f_mfma.log 2>&1; echo "pmc mfma $?"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/r05f_sq -o p -- python3 $R/bench.py --steps 3 --warmup 1 $Q > $O/r05f_sq.log 2>&1; echo "pmc sq $?"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/r05f_sq2 -o p -- python3 $R/bench.py --steps 3 --warmup 1 $Q > $O/r05f_sq2.log 2>&1; echo "pmc sq2 $?"
cd $R
f() { find $O/$1 -name "*$2" | head -1; }
python3 tools/pmc_to_json.py $(f r05f_pmc_FETCH_SIZE counter_collection.csv) $(f r05f_pmc_WRITE_SIZE counter_collection.csv) $O/pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of \`bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-int8-extra --no-dispatcher --no-two-lanes\`, r05 final build" > /dev/null
python3 tools/pmc_to_json.py $(f r05f_pmc_i8_FETCH_SIZE counter_collection.csv) $(f r05f_pmc_i8_WRITE_SIZE counter_collection.csv) $O/pmc_traffic_int8.json "the same passes of \`bench.py --int8 ...\`, r05 final build" > /dev/null
python3 tools/mfma_util.py $(f r05f_mfma counter_collection.csv) $(f r05f_mfma kernel_trace.csv) $O/r05_mfma_util.json > $O/r05_mfma_util.txt
python3 tools/sq_breakdown.py $(f r05f_sq counter_collection.csv) $(f r05f_sq kernel_trace.csv) > $O/r05_sq_breakdown.txt
python3 tools/sq_breakdown.py $(f r05f_sq2 counter_collection.csv) $(f r05f_sq2 kernel_trace.csv) > $O/r05_sq_lds_breakdown.txt
cp $(f r05f_kt kernel_stats.csv) $O/r05_kernel_stats.csv; cp $(f r05f_kt_i8 kernel_stats.csv) $O/r05_int8_kernel_stats.csv
cp $(f r05f_kt2 kernel_stats.csv) $O/r05_kernel_stats_2lanes.csv; cp $O/r05f_kt2.json $O/r05_bench_under_rocprof_2lanes.json
cp $O/r05f_kt.json $O/r05_bench_under_rocprof.json       # the bench line of the run r05_kernel_stats.csv was taken from (compare roofline.avg_launch_ms)
rm -rf $O/r05f_*                                          # the raw traces: gpurun copies back at most 64 MiB
head -5 $O/r05_mfma_util.txt; head -6 $O/r05_sq_breakdown.txt
