#!/bin/bash
# A/B of le_rows.hip compile-time knobs on ONE box (boxes of the pool differ by +-5 %): builds a scratch copy of the library per
# variant and times the fused LE layers.  usage: bash tools/ab_rows.sh "" "-DROWS_PIN_H=2" "-DROWS_PIN_F=0 -DROWS_AHEAD_F=6" ...
set -e -o pipefail
for defs in "$@"; do
  rm -rf /tmp/abbuild && mkdir -p /tmp/abbuild && cp -r hdr-realtime-video-pipeline_amd include tools tests oracle bench.py /tmp/abbuild/
  (cd /tmp/abbuild/hdr-realtime-video-pipeline_amd/csrc && touch le_rows.hip && make EXTRA="$defs" -j8 > /tmp/abbuild/make.log 2>&1) || { tail -5 /tmp/abbuild/make.log; continue; }
  (cd /tmp/abbuild && python bench.py --steps 20 --warmup 5 --layers --no-hg --no-cpu-baseline --no-dispatcher --no-int8-extra 2>&1 > /dev/null \
     | grep -E "\[layer\] LE\.(head|tail|recon_trunk1)" | awk -v d="$defs" '{printf "[%s] %s %s ms | ", d, $2, $4} END {print ""}')
done
