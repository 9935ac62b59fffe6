#!/bin/bash
# kernel trace of the bench -> gap table (+ what a gradient-penalty iteration adds) and the per-kernel summary
set -e -o pipefail
tag=${1:-r3_xx}
out=gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/${tag}_kta -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $out/${tag}_bench_under_rocprof.log 2> $out/${tag}_rocprof.err
python3 tools/gap_table.py "$(ls /tmp/${tag}_kta/*/*kernel_trace.csv | head -1)" 5 > $out/${tag}_gap_table.txt
python3 tools/prof_summary.py "$(ls /tmp/${tag}_kta/*/*kernel_trace.csv | head -1)" 25 > $out/${tag}_kernel_stats.txt
echo "trace done"
