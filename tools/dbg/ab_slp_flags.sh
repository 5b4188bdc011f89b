#!/bin/bash
# The library with and without the SLP / loop vectorisers on ONE box: frames/s (one lane) and the LE / HG times of bench.py --layers.
# usage (on the GPU box, from the repo root): bash tools/dbg/ab_slp_flags.sh
R=$(pwd)
rm -rf /tmp/abslp; mkdir -p /tmp/abslp; cp -r $R/hdr-realtime-video-pipeline_amd $R/include $R/tools $R/tests $R/oracle $R/bench.py /tmp/abslp/
( cd /tmp/abslp/hdr-realtime-video-pipeline_amd/csrc && sed -i 's/^CXXFLAGS += -fno-slp-vectorize -fno-vectorize$/# (vectorisers on)/' Makefile && make clean > /dev/null && make -j16 > /tmp/abslp/make.log 2>&1 || tail -3 /tmp/abslp/make.log )
Q="--steps 30 --warmup 5 --no-cpu-baseline --no-dispatcher --no-int8-extra --no-two-lanes --no-latency-tail"
for rep in 1 2; do
  for w in $R /tmp/abslp; do
    ( cd $w && python bench.py $Q 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w'.replace('/tmp/abslp','vectorisers ON ').replace('$R','vectorisers OFF'), d['value'], 'frames/s; infer', d['roofline']['infer_ms_profiled'], 'ms')" )
  done
done
