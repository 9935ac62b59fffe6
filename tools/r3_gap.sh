#!/bin/bash
# Round 3, step 1: where does the GPU idle?  bench with blocking / asynchronous metrics, kernel traces of both -> gap tables.
set -e -o pipefail
tag=${1:-r3_01}
out=gpurun_out
export TMPDIR=/tmp
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --sync-metrics > $out/${tag}_bench_sync.log 2> $out/${tag}_bench_sync.err
echo "sync bench done"; python3 tools/bench_summary.py $out/${tag}_bench_sync.log || true
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $out/${tag}_bench_async.log 2> $out/${tag}_bench_async.err
echo "async bench done"; python3 tools/bench_summary.py $out/${tag}_bench_async.log || true
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/${tag}_kts -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --sync-metrics > $out/${tag}_bench_sync_under_rocprof.log 2> $out/${tag}_rocprof_s.err
python3 tools/gap_table.py "$(ls /tmp/${tag}_kts/*/*kernel_trace.csv | head -1)" 5 > $out/${tag}_gap_table_sync.txt
echo "sync trace done"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/${tag}_kta -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $out/${tag}_bench_async_under_rocprof.log 2> $out/${tag}_rocprof_a.err
python3 tools/gap_table.py "$(ls /tmp/${tag}_kta/*/*kernel_trace.csv | head -1)" 5 > $out/${tag}_gap_table_async.txt
python3 tools/prof_summary.py "$(ls /tmp/${tag}_kta/*/*kernel_trace.csv | head -1)" 25 > $out/${tag}_kernel_stats.txt
echo "async trace done"
