#!/bin/bash
# as tools/ab_files.sh, for the INT8 configuration (bench.py --int8 [args])
R=$GRAFT_REPO_ROOT; cd $R
files=$1; pat=${2:-.}; shift; shift
for v in new old new2; do
  rm -rf /tmp/ab_$v && mkdir -p /tmp/ab_$v && cp -r hdr-realtime-video-pipeline_amd include tools tests bench.py oracle BASELINE.json /tmp/ab_$v/
  if [ $v = old ]; then for f in $files; do cp tools/build/old/$f /tmp/ab_$v/hdr-realtime-video-pipeline_amd/csrc/$f; done; touch /tmp/ab_$v/hdr-realtime-video-pipeline_amd/csrc/*.hip; fi
  (cd /tmp/ab_$v/hdr-realtime-video-pipeline_amd/csrc && for f in $files; do touch $f; done && make -j8 2>&1 | grep -E "error" )
  (cd /tmp/ab_$v && python bench.py --int8 --steps 20 --warmup 5 --layers --no-cpu-baseline --no-int8-extra --no-dispatcher "$@" 2> $R/gpurun_out/ab_$v.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['p50_ms'])")
  grep "^\[kernel\]" $R/gpurun_out/ab_$v.err | grep -E "$pat" | cut -c1-110
done
