#!/bin/bash
# same as r3_ab.sh, plus a kernel trace of the first variant (per-kernel averages for the named patterns)
set -e -o pipefail
tag=$1; shift
pat=$1; shift
bash tools/r3_ab.sh $tag "$@"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/${tag}_kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/${tag}_kt -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-extra $1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/${tag}_kt -name '*kernel_stats.csv' | head -1)
python3 - "$f" "$pat" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2])
for r in rows:
    if pat.search(r['Name']):
        print(f"{float(r['TotalDurationNs'])/1e6:9.3f} ms {int(r['Calls']):6d} calls {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:110]}")
PY
