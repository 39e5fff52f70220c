#!/bin/bash
# env-tunable sweep on the c2 bench (graph-replay ms/step); usage: tools/sweep.sh "VAR=val" "VAR2=val" ...
for kv in "BASE=1" "$@"; do
  ms=$(env "ITCV_$kv" timeout -k 10 150 python3 bench.py --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'])")
  echo "$kv $ms"
done
