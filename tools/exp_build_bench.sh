#!/bin/bash
# Diagnostic: build csrc with EXTRA="$1" in a scratch copy and print the per-layer table of a 4K frame (results are not checked).
set -e -o pipefail
rm -rf /tmp/expbuild && mkdir -p /tmp/expbuild && cp -r hdr-realtime-video-pipeline_amd include tools tests bench.py oracle BASELINE.json /tmp/expbuild/ 2>/dev/null || true
cd /tmp/expbuild/hdr-realtime-video-pipeline_amd/csrc && rm -rf build ../lib/*.so && (make EXTRA="$1" -j8 2>&1 | grep -v warning | tail -1)
cd /tmp/expbuild && python bench.py --steps 30 --warmup 5 --layers --no-cpu-baseline --no-int8-extra 2>&1 | grep "\[layer\]\|\"value\"" | grep "${2:-layer}"
