#!/bin/bash
# full GPU suite with a heartbeat file (the box's silence detector kills a run that writes nothing for 3 minutes)
out=gpurun_out
( while true; do date >> $out/hb.txt; sleep 20; done ) &
HB=$!
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 150 --timeout-method thread > $out/suite.log 2>&1
rc=$?
kill $HB
tail -6 $out/suite.log
exit $rc
