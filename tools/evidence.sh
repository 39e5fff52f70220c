#!/bin/bash
# Round evidence on the GPU box: per workload the two --pmc passes (HBM bytes per launch), the kernel-trace summary and
# the bench line.  Run as: gpurun -- 'bash tools/evidence.sh'; then copy gpurun_out/ev/r03_* into profiles/.
# The PMC passes run WITHOUT any trace domain beside the kernel trace (gpurun refuses other combinations).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ev
MATH=${MATH:-f16x3}
PMC_ARGS=""
for c in c2 c3 c5; do
  ARGS="--config $c --math $MATH --steps 2 --warmup 1 --no-graph --no-modes --no-cpu-baseline"
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/ev/pmc_fetch_$c -o f -- python3 bench.py $ARGS > gpurun_out/ev/pmc_fetch_$c.json 2> gpurun_out/ev/pmc_fetch_$c.err
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/ev/pmc_write_$c -o w -- python3 bench.py $ARGS > gpurun_out/ev/pmc_write_$c.json 2> gpurun_out/ev/pmc_write_$c.err
  PMC_ARGS="$PMC_ARGS $c $(find gpurun_out/ev/pmc_fetch_$c -name '*counter_collection.csv') $(find gpurun_out/ev/pmc_write_$c -name '*counter_collection.csv')"
  echo "pmc $c done"
done
python tools/pmc_summary.py $MATH $PMC_ARGS > gpurun_out/ev/r03_pmc_traffic.json 2> gpurun_out/ev/pmc_summary.log
cp gpurun_out/ev/r03_pmc_traffic.json profiles/r03_pmc_traffic.json
rm -rf gpurun_out/ev/pmc_fetch_c* gpurun_out/ev/pmc_write_c*
for c in c2 c3 c5; do
  S="--steps 20 --warmup 5"; [ $c != c2 ] && S="--steps 6 --warmup 3"
  timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/ev/prof_$c -o p -- python3 bench.py --config $c --no-cpu-baseline --no-modes $S > gpurun_out/ev/prof_$c.json 2> gpurun_out/ev/prof_$c.err
  python tools/rocpd_stats.py gpurun_out/ev/prof_$c/p_results.db > gpurun_out/ev/r03_${c}_kernel_stats.csv 2> gpurun_out/ev/prof_${c}_span.txt
  rm -rf gpurun_out/ev/prof_$c
  echo "prof $c done"
done
timeout -k 10 900 python bench.py > gpurun_out/ev/r03_bench_c2.json 2> gpurun_out/ev/bench_c2.err
echo "bench c2 done"
timeout -k 10 600 python bench.py --config c3 --no-modes > gpurun_out/ev/r03_bench_c3.json 2> gpurun_out/ev/bench_c3.err
echo "bench c3 done"
timeout -k 10 600 python bench.py --config c5 --no-modes > gpurun_out/ev/r03_bench_c5.json 2> gpurun_out/ev/bench_c5.err
echo "bench c5 done"
