#!/bin/bash
# SQ wave-cycle breakdown + LDS counters of the le_rows.hip kernels (run from the repo root on the GPU box)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-rows}
cd /tmp && export TMPDIR=/tmp
B="--steps 3 --warmup 1 --no-hg --no-cpu-baseline --no-int8-extra --no-dispatcher"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/sq_$tag -o p -- python3 $R/bench.py $B > $O/sq_$tag.log 2>&1
echo "pmc sq $tag exit $?"
f=$(find $O/sq_$tag -name '*counter_collection.csv' | head -1); k=$(find $O/sq_$tag -name '*kernel_trace.csv' | head -1)
python3 $R/tools/sq_breakdown.py $f $k rows > $O/sq_$tag.txt; cat $O/sq_$tag.txt
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/sq2_$tag -o p -- python3 $R/bench.py $B > $O/sq2_$tag.log 2>&1
echo "pmc sq2 $tag exit $?"
f=$(find $O/sq2_$tag -name '*counter_collection.csv' | head -1); k=$(find $O/sq2_$tag -name '*kernel_trace.csv' | head -1)
python3 $R/tools/sq_breakdown.py $f $k rows > $O/sq2_$tag.txt; cat $O/sq2_$tag.txt
