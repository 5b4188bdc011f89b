#!/bin/bash
# SQ wave-cycle breakdown + HBM/L2-miss traffic of the HG conv kernels (run from the repo root on the GPU box)
# usage: tools/r03_pmc_sq.sh <tag> [ENV=VALUE ...]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/sq_$tag -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-int8-extra --no-dispatcher > $O/sq_$tag.log 2>&1
echo "pmc sq $tag exit $?"
f=$(find $O/sq_$tag -name '*counter_collection.csv' | head -1); k=$(find $O/sq_$tag -name '*kernel_trace.csv' | head -1)
python3 $R/tools/sq_breakdown.py $f $k conv_p > $O/sq_$tag.txt; cat $O/sq_$tag.txt
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/tr_${tag}_$ctr -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-int8-extra --no-dispatcher > $O/tr_${tag}_$ctr.log 2>&1
  f=$(find $O/tr_${tag}_$ctr -name '*counter_collection.csv' | head -1)
  python3 - "$f" $ctr <<'PY'
import csv, sys, collections
a = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2] and "conv_p" in r["Kernel_Name"]:
        n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        a[n][0] += float(r["Counter_Value"]); a[n][1] += 1
for n, (v, c) in sorted(a.items()):
    print(f"{sys.argv[2]} {n:32s} launches={c:4d} per-launch={v / c:12.1f} (KB as reported)")
PY
done
