#!/bin/bash
set -e -o pipefail
tag=${1:-r3_xx}
out=gpurun_out
export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d /tmp/${tag}_sq -- python3 tools/run_hot_kernels.py > $out/${tag}_sq.err 2>&1
python3 tools/pmc_table.py /tmp/${tag}_sq > $out/${tag}_pmc_sq_kernels.txt
echo "sq done"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 150 rocprofv3 --pmc $c --output-format csv -d /tmp/${tag}_$c -- python3 tools/run_hot_kernels.py > $out/${tag}_$c.err 2>&1
  python3 tools/pmc_table.py /tmp/${tag}_$c > $out/${tag}_pmc_${c}_kernels.txt
  echo "$c done"
done
