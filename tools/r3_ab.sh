#!/bin/bash
# A/B of bench.py variants: bash tools/r3_ab.sh <tag> "<flags A>" "<flags B>" ...
set -e -o pipefail
tag=$1; shift
out=gpurun_out
i=0
for flags in "$@"; do
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra $flags > $out/${tag}_v$i.log 2> $out/${tag}_v$i.err
  echo "variant $i [$flags]"; python3 tools/bench_summary.py $out/${tag}_v$i.log
  i=$((i+1))
done
