#!/bin/bash
# Diagnostic: STAMP build in a scratch copy of csrc (the shipped library stays untouched), per-phase cycle shares of one layer.
set -e -o pipefail
L=${1:-LE.HR_conv1}
rm -rf /tmp/stampbuild && mkdir -p /tmp/stampbuild && cp -r hdr-realtime-video-pipeline_amd include tools tests /tmp/stampbuild/
cd /tmp/stampbuild/hdr-realtime-video-pipeline_amd/csrc && rm -rf build ../lib/*.so && (make STAMP=1 -j8 2>&1 | grep -v warning | tail -3)
cd /tmp/stampbuild && python tools/${TOOL:-stamp_conv32p.py} $L
