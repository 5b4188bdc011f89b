#!/bin/bash
# usage (on the GPU box, repo root): tools/ab_build.sh <tag> <file.hip> "<EXTRA flags>" "<regex over [kernel]/[layer] lines>"
# rebuilds one file with extra compiler flags, runs the layer bench, prints the matching lines
tag=$1; f=$2; extra=$3; pat=${4:-"\[kernel\]"}
cd $GRAFT_REPO_ROOT/hdr-realtime-video-pipeline_amd/csrc && touch $f && make EXTRA="$extra" > $GRAFT_REPO_ROOT/gpurun_out/build_$tag.log 2>&1
cd $GRAFT_REPO_ROOT && python bench.py --layers --steps 20 --no-cpu-baseline 2> gpurun_out/layers_$tag.txt | cut -c1-200
grep -E "$pat" gpurun_out/layers_$tag.txt | awk -v t=$tag '{print t, $2, $3, $4, $5, $6}'
