#!/bin/bash
# conv32s.hip built (on the GPU box, scratch copy) with a source edit of tools/dbg/conv32s_edit.py; prints the int8 twin difference.
R=$(pwd)
for e in "$@"; do
  rm -rf /tmp/slpb; mkdir -p /tmp/slpb; cp -r $R/hdr-realtime-video-pipeline_amd $R/include $R/tools $R/tests /tmp/slpb/
  ( cd /tmp/slpb/hdr-realtime-video-pipeline_amd/csrc && python $R/tools/dbg/conv32s_edit.py conv32s.hip $e && make -j16 > /tmp/slpb/make.log 2>&1 || tail -3 /tmp/slpb/make.log )
  echo "== conv32s.hip edit: $e"
  ( cd /tmp/slpb && python tools/dbg/twin_repeat.py int8 2>&1 | grep "^le\.fea0" | cut -c1-110 )
done
