#!/bin/bash
# A/B of bench.py --config 3 variants: bash tools/r3_ab3.sh <tag> "<flags A>" "<flags B>" ...
set -e -o pipefail
tag=$1; shift
out=gpurun_out
i=0
for flags in "$@"; do
  timeout -k 10 300 python3 bench.py --config 3 $flags > $out/${tag}_v$i.log 2> $out/${tag}_v$i.err
  echo "variant $i [$flags]"; python3 - $out/${tag}_v$i.log <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        j = json.loads(l)
        print('  ms/step', round(j['ms_per_step'], 1), 'traj/s', round(j['value']), 'frac', j.get('roofline', {}).get('frac'))
        k = j.get('roofline', {}).get('all_kernels_ms_per_step') or {}
        print('  ', {n: round(v, 1) for n, v in sorted(k.items(), key=lambda kv: -kv[1])[:12]})
PY
  i=$((i+1))
done
