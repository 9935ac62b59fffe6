#!/bin/bash
set -e -o pipefail
tag=${1:-tl}
out=gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/${tag}_kta -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extra > $out/${tag}_bench_under_rocprof.log 2> $out/${tag}_rocprof.err
f="$(ls /tmp/${tag}_kta/*/*kernel_trace.csv | head -1)"
python3 tools/timeline.py "$f" 9 20 > $out/${tag}_timeline_plain.txt
python3 tools/timeline.py "$f" 8 20 > $out/${tag}_timeline_gp.txt
python3 tools/timeline.py "$f" 9 0 > $out/${tag}_timeline_plain_all.txt
python3 tools/gap_table.py "$f" 5 > $out/${tag}_gap_table.txt
python3 tools/prof_summary.py "$f" 16 > $out/${tag}_kernel_stats.txt
echo "timeline done"
