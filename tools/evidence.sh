#!/bin/bash
# Round evidence on the GPU box: the two --pmc passes (HBM bytes per launch), the kernel-trace summary and the default
# bench line of the c2 workload.  Run as: gpurun -- 'bash tools/evidence.sh'; then copy gpurun_out/ev/r02_* into profiles/.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ev
ARGS="--steps 2 --warmup 1 --no-graph --no-modes --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/ev/pmc_fetch -o f -- python3 bench.py $ARGS > gpurun_out/ev/pmc_fetch.json 2> gpurun_out/ev/pmc_fetch.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/ev/pmc_write -o w -- python3 bench.py $ARGS > gpurun_out/ev/pmc_write.json 2> gpurun_out/ev/pmc_write.err
python tools/pmc_summary.py $(find gpurun_out/ev/pmc_fetch -name '*counter_collection.csv') $(find gpurun_out/ev/pmc_write -name '*counter_collection.csv') > gpurun_out/ev/r02_pmc_traffic.json 2> gpurun_out/ev/pmc_summary.log
cp gpurun_out/ev/r02_pmc_traffic.json profiles/r02_pmc_traffic.json
echo "pmc done"
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/ev/prof_c2 -o p -- python3 bench.py --no-cpu-baseline --no-modes > gpurun_out/ev/prof_c2.json 2> gpurun_out/ev/prof_c2.err
python tools/rocpd_stats.py gpurun_out/ev/prof_c2/p_results.db > gpurun_out/ev/r02_c2_kernel_stats.csv 2> gpurun_out/ev/prof_c2_span.txt
rm -rf gpurun_out/ev/prof_c2
echo "prof c2 done"
timeout -k 10 600 python bench.py > gpurun_out/ev/r02_bench_c2.json 2> gpurun_out/ev/bench_c2.err
echo "bench c2 done"
timeout -k 10 600 python bench.py --config c3 --no-modes > gpurun_out/ev/r02_bench_c3.json 2> gpurun_out/ev/bench_c3.err
echo "bench c3 done"
timeout -k 10 600 python bench.py --config c5 --no-modes > gpurun_out/ev/r02_bench_c5.json 2> gpurun_out/ev/bench_c5.err
echo "bench c5 done"
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/ev/prof_c3 -o p -- python3 bench.py --config c3 --no-cpu-baseline --no-modes --steps 6 --warmup 3 > gpurun_out/ev/prof_c3.json 2> gpurun_out/ev/prof_c3.err
python tools/rocpd_stats.py gpurun_out/ev/prof_c3/p_results.db > gpurun_out/ev/r02_c3_kernel_stats.csv 2> gpurun_out/ev/prof_c3_span.txt
rm -rf gpurun_out/ev/prof_c3
echo "prof c3 done"
