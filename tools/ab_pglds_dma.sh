#!/bin/bash
# one box: conv_pglds with global_load_lds (round-2 form) against the buffer_load ... lds form; HDRTV_VARIANTS=prw=0 so that every HG layer runs it
R=$GRAFT_REPO_ROOT; cd $R
for v in buffer global; do
  rm -rf /tmp/ab_$v && mkdir -p /tmp/ab_$v && cp -r hdr-realtime-video-pipeline_amd include tools tests bench.py oracle BASELINE.json /tmp/ab_$v/
  if [ $v = global ]; then cp tools/build/conv3x3_pglds_globaldma.hip /tmp/ab_$v/hdr-realtime-video-pipeline_amd/csrc/conv3x3_pglds.hip; fi
  (cd /tmp/ab_$v/hdr-realtime-video-pipeline_amd/csrc && touch conv3x3_pglds.hip && make -j8 2>&1 | grep -E "error" )
  (cd /tmp/ab_$v && HDRTV_VARIANTS=prw=0 python bench.py --steps 20 --warmup 5 --layers --no-cpu-baseline --no-int8-extra --no-dispatcher 2> $R/gpurun_out/ab_pglds_$v.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['p50_ms'])")
  grep "^\[kernel\] conv_pglds" $R/gpurun_out/ab_pglds_$v.err | cut -c1-110
done
