#!/bin/bash
# rocprofv3 kernel-trace summaries of the c2 step per conv arithmetic (gpurun -- 'bash tools/prof_modes.sh f16x3 bf16x3')
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pm
for m in "$@"; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/pm/prof_$m -o p -- python3 bench.py --math $m --no-cpu-baseline --no-modes --steps 20 --warmup 3 > gpurun_out/pm/bench_$m.json 2> gpurun_out/pm/bench_$m.err || exit 1
  python tools/rocpd_stats.py gpurun_out/pm/prof_$m/p_results.db > gpurun_out/pm/kernel_stats_$m.csv 2> gpurun_out/pm/span_$m.txt
  rm -rf gpurun_out/pm/prof_$m
  echo "prof $m done"
done
