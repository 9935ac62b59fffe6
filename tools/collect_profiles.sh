#!/bin/bash
# Collects the per-round measurement artefacts on the GPU box (run through gpurun from the repository root):
#   bash tools/collect_profiles.sh r2_02
# Every GPU step has its own time limit and the chain stops at the first failure.  Outputs land in gpurun_out/<tag>_*;
# copy the ones to keep into profiles/.
set -e -o pipefail
tag=${1:-rX}
out=gpurun_out
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py > $out/${tag}_bench_line.log 2> $out/${tag}_bench_line.err
echo "bench done"
timeout -k 10 200 python3 bench.py --config 5 --no-cpu-baseline > $out/${tag}_bench_config5.log 2> $out/${tag}_bench_config5.err
echo "config 5 done"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/${tag}_kt -- python3 bench.py --no-cpu-baseline > $out/${tag}_bench_line_under_rocprof.log 2> $out/${tag}_rocprof.err
python3 tools/prof_summary.py "$(ls /tmp/${tag}_kt/*/*kernel_trace.csv | head -1)" 16 > $out/${tag}_kernel_stats.txt
echo "kernel trace done"
timeout -k 10 300 python3 tools/bench_kernels.py > $out/${tag}_per_kernel_roofline.txt 2> $out/${tag}_per_kernel.err
echo "per-kernel done"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d /tmp/${tag}_pmc_$c -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 4 > /dev/null 2> $out/${tag}_pmc_$c.err
  python3 tools/pmc_table.py /tmp/${tag}_pmc_$c > $out/${tag}_pmc_bench_$c.txt
  echo "pmc $c (bench) done"
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d /tmp/${tag}_pmck_$c -- python3 tools/bench_kernels.py > /dev/null 2> $out/${tag}_pmck_$c.err
  python3 tools/pmc_table.py /tmp/${tag}_pmck_$c > $out/${tag}_pmc_kernels_$c.txt
  echo "pmc $c (kernels) done"
done
# config 3 (bf16, H = 512): bench line + kernel-trace summary + the forward step's fabric traffic
timeout -k 10 300 python3 bench.py --config 3 --no-cpu-baseline > $out/${tag}_bench_config3.log 2> $out/${tag}_bench_config3.err
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/${tag}_kt3 -- python3 bench.py --config 3 --no-cpu-baseline > $out/${tag}_bench_config3_under_rocprof.log 2> $out/${tag}_rocprof3.err
python3 tools/prof_summary.py "$(ls /tmp/${tag}_kt3/*/*kernel_trace.csv | head -1)" 9 > $out/${tag}_kernel_stats_config3.txt
echo "config 3 done"
