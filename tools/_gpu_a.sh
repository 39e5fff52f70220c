cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/b/prof_c2 -o p -- python3 bench.py --no-cpu-baseline --no-modes --steps 20 --warmup 5 > gpurun_out/b/prof_c2.json 2> gpurun_out/b/prof_c2.err && \
python tools/rocpd_stats.py gpurun_out/b/prof_c2/p_results.db > gpurun_out/b/c2_stats.csv 2> gpurun_out/b/c2_span.txt && rm -rf gpurun_out/b/prof_c2
grep -h "wgrad5\|conv_fwd_bf16p3_kernel<6, 64, false" gpurun_out/b/c2_stats.csv | cut -c1-70,200-300
