#!/bin/bash
# LDS counters of the le_rows.hip kernels (GPU box, repo root): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per kernel, no-HG bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r5rows}
cd /tmp && export TMPDIR=/tmp
B="--steps 3 --warmup 1 --no-hg --no-cpu-baseline --no-int8-extra --no-dispatcher"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/sq2_$tag -o p -- python3 $R/bench.py $B > $O/sq2_$tag.log 2>&1
echo "pmc sq2 $tag exit $?"
f=$(find $O/sq2_$tag -name '*counter_collection.csv' | head -1); k=$(find $O/sq2_$tag -name '*kernel_trace.csv' | head -1)
python3 $R/tools/sq_breakdown.py $f $k > $O/sq2_$tag.txt; grep -E "rows|cond_trunk|preg" $O/sq2_$tag.txt
rm -rf $O/sq2_$tag
