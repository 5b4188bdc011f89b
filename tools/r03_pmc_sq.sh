#!/bin/bash
# SQ wave-cycle breakdown of the HG conv kernels, both schedules (run from the repo root on the GPU box)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in 0 2; do
  HDRTV_PRW=$v rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/sq_prw$v -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-int8-extra --no-dispatcher > $O/sq_prw$v.log 2>&1
  echo "pmc prw=$v exit $?"
  f=$(find $O/sq_prw$v -name '*counter_collection.csv' | head -1); k=$(find $O/sq_prw$v -name '*kernel_trace.csv' | head -1)
  python3 $R/tools/sq_breakdown.py $f $k conv_p > $O/sq_prw$v.txt; cat $O/sq_prw$v.txt
done
