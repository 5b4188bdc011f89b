#!/bin/bash
# usage (GPU box, repo root): tools/kstat.sh "<grep -i pattern over kernel names>" [bench args]  -- rocprofv3 kernel stats of a short bench run
R=$GRAFT_REPO_ROOT; pat=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/kstat
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kstat -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-int8-extra --no-dispatcher "$@" > $R/gpurun_out/kstat.json 2> $R/gpurun_out/kstat.err || { echo "bench failed"; tail -5 $R/gpurun_out/kstat.err; exit 1; }
f=$(find $R/gpurun_out/kstat -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] || { echo "no kernel_stats.csv"; exit 1; }
python3 - "$f" "$pat" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"], re.I):
        print(f'{r["Name"][:70]:70s} n={r["Calls"]:>5s} avg_us={float(r["AverageNs"]) / 1000:9.1f}')
PY
python3 -c "import json; d=json.loads(open('$R/gpurun_out/kstat.json').read().strip().splitlines()[-1]); print('fps', d['value'], 'p50', d['p50_ms'])"
