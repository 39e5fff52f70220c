#!/bin/bash
# Repeat tools/ddp_graph_probe.py with and without the pre-capture watchdog drain; stops at the first timeout.
for drain in 0 0.35; do
  for i in 1 2 3 4 5 6 7 8; do
    ITCV_DDP_DRAIN=$drain MASTER_PORT=$((29540 + i)) timeout -k 10 120 python tools/ddp_graph_probe.py > gpurun_out/probe_${drain}_$i.log 2>&1
    rc=$?
    echo "drain=$drain run=$i rc=$rc $(grep -c 'watchdog thread terminated' gpurun_out/probe_${drain}_$i.log)"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  done
done
