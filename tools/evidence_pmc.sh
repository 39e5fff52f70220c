#!/bin/bash
# The two --pmc passes + fold + default bench line only (see tools/evidence.sh for the full set).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ev
ARGS="--steps 2 --warmup 1 --no-graph --no-modes --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/ev/pmc_fetch -o f -- python3 bench.py $ARGS > gpurun_out/ev/pmc_fetch.json 2> gpurun_out/ev/pmc_fetch.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/ev/pmc_write -o w -- python3 bench.py $ARGS > gpurun_out/ev/pmc_write.json 2> gpurun_out/ev/pmc_write.err
python tools/pmc_summary.py $(find gpurun_out/ev/pmc_fetch -name '*counter_collection.csv') $(find gpurun_out/ev/pmc_write -name '*counter_collection.csv') > gpurun_out/ev/r02_pmc_traffic.json 2> gpurun_out/ev/pmc_summary.log
cp gpurun_out/ev/r02_pmc_traffic.json profiles/r02_pmc_traffic.json
echo "pmc done"
timeout -k 10 600 python bench.py > gpurun_out/ev/r02_bench_c2.json 2> gpurun_out/ev/bench_c2.err
echo "bench c2 done"
