#!/bin/bash
# as tools/ab_flags.sh, for `bench.py --int8` (native full recipe)
set -e -o pipefail
pat="$1"; shift
for defs in "$@"; do
  rm -rf /tmp/abbuild && mkdir -p /tmp/abbuild && cp -r hdr-realtime-video-pipeline_amd include tools tests oracle bench.py profiles /tmp/abbuild/
  (cd /tmp/abbuild/hdr-realtime-video-pipeline_amd/csrc && touch *.hip && make EXTRA="$defs" -j16 > /tmp/abbuild/make.log 2>&1) || { tail -5 /tmp/abbuild/make.log; continue; }
  (cd /tmp/abbuild && python bench.py --int8 --steps 20 --warmup 5 --layers --no-cpu-baseline --no-dispatcher 2>&1 > /tmp/abbuild/line.json \
     | grep -E "\[kernel\] ($pat)" | awk -v d="$defs" '{printf "[%s] %s %s ms | ", d, $2, $6} END {print ""}'; python3 -c "import json;print('   frames/s', json.loads(open('/tmp/abbuild/line.json').read().strip().splitlines()[-1])['value'])")
done
