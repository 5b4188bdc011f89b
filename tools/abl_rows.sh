#!/bin/bash
# Diagnostic: what is a le_rows.hip kernel's time made of?  Builds scratch copies of the library with parts of the kernel left
# out (RB_ABL bits: 1 no LDS-DMA, 2 no stores, 4 no conv MFMAs, 8 no SFT passes, 16 no barrier) and times the layer.
# usage: bash tools/abl_rows.sh "LE.recon_trunk1" 0 1 2 3 4 8 12 16 ...
set -e -o pipefail
L=$1; shift
for m in "$@"; do
  rm -rf /tmp/ablbuild && mkdir -p /tmp/ablbuild && cp -r hdr-realtime-video-pipeline_amd include tools tests oracle bench.py /tmp/ablbuild/
  (cd /tmp/ablbuild/hdr-realtime-video-pipeline_amd/csrc && touch le_rows.hip && make EXTRA=-DRB_ABL=$m -j8 > /dev/null 2>&1)
  (cd /tmp/ablbuild && python bench.py --steps 10 --warmup 3 --layers --no-hg --no-cpu-baseline --no-dispatcher --no-int8-extra 2>&1 > /dev/null | grep "$L" | sed "s/^/ABL=$m /")
done
