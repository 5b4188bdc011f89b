#!/bin/bash
# usage: tools/ab_env.sh VAR v1 v2 ... : bench each setting, print kernel lines matching $AB_GREP and the fps
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --steps 8 --warmup 3 --layers --no-cpu-baseline > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  echo "== $var=$v fps=$(python -c "import json;print(json.load(open('gpurun_out/ab_'+'$v'+'.json'))['value'])")"
  grep '^\[kernel\]' gpurun_out/ab_$v.err | grep -E "${AB_GREP:-.}" | cut -c1-100
done
