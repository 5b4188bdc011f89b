#!/bin/bash
# usage (on the GPU box, repo root): tools/ab_build.sh <tag> <file.hip> "<EXTRA flags>"  -> rebuilds one file with flags, runs the layer bench
tag=$1; f=$2; extra=$3
cd $GRAFT_REPO_ROOT/hdr-realtime-video-pipeline_amd/csrc && touch $f && make EXTRA="$extra" > $GRAFT_REPO_ROOT/gpurun_out/build_$tag.log 2>&1
cd $GRAFT_REPO_ROOT && python bench.py --layers --steps 20 --no-cpu-baseline 2> gpurun_out/layers_$tag.txt | cut -c1-200
grep -E "\[layer\] hg\.(conv2|conv3_1|conv4_1|conv5_1|Up_conv4|Up_conv5) " gpurun_out/layers_$tag.txt | awk -v t=$tag '{print t, $2, $4, $6}'
