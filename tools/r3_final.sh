#!/bin/bash
# Round-3 record: the driver's command (bench line with cpu_baseline + extra_configs), the kernel trace of the same step
# (gap table + per-kernel summary), the penalty pass in isolation.
set -e -o pipefail
tag=${1:-r3_05}
out=gpurun_out
export TMPDIR=/tmp
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_line.log 2> $out/${tag}_bench_line.err
echo "bench done"; python3 tools/bench_summary.py $out/${tag}_bench_line.log
bash tools/r3_trace.sh ${tag}
timeout -k 10 120 python3 tools/bench_gp.py > $out/${tag}_gp_pass.txt 2>&1
echo "gp done"
