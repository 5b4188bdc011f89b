#!/bin/bash
# Which file's loss of the SLP vectoriser makes the int8 row kernels differ from the per-layer int8 kernels?  For each argument (a list
# of csrc files without .hip), a scratch copy of the library is built ON THE GPU BOX with the vectorisers back on for those files only,
# and tools/dbg/twin_repeat.py prints how many values of le.fea0 differ between the two forms (MODE=fp16: the cond2 / cond3 tails).
# usage: bash tools/dbg/slp_bisect.sh "" le_rows_i8 conv32s le_hg_misc conv_q8
R=$(pwd)
for f in "$@"; do
  rm -rf /tmp/slpb; mkdir -p /tmp/slpb; cp -r $R/hdr-realtime-video-pipeline_amd $R/include $R/tools $R/tests /tmp/slpb/
  ( cd /tmp/slpb/hdr-realtime-video-pipeline_amd/csrc
    for one in $f; do printf '\n$(BUILD)/%s.o: CXXFLAGS += -fslp-vectorize -fvectorize\n' "$one" >> Makefile; touch $one.hip; done
    make -j16 > /tmp/slpb/make.log 2>&1 || tail -3 /tmp/slpb/make.log )
  echo "== vectorisers on for: [$f]"
  ( cd /tmp/slpb && python tools/dbg/twin_repeat.py ${MODE:-int8} 2>&1 | grep "^le\." | cut -c1-110 )
done
