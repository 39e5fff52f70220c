#!/bin/bash
# Repeat the RCCL graph-capture test (diagnostic for the watchdog / capture race); stops at the first timeout.
for i in 1 2 3 4 5 6; do
  timeout -k 10 300 python -m pytest tests/test_ddp.py -x -q -m gpu -k graph_capture > gpurun_out/ddp_loop_$i.log 2>&1
  rc=$?
  echo "run=$i rc=$rc $(tail -1 gpurun_out/ddp_loop_$i.log)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
