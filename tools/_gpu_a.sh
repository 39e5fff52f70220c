cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/b/tests.log 2>&1 || { tail -30 gpurun_out/b/tests.log; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/b/prof_c2 -o p -- python3 bench.py --no-cpu-baseline --no-modes --steps 20 --warmup 5 > gpurun_out/b/prof_c2.json 2> gpurun_out/b/prof_c2.err && \
python tools/rocpd_stats.py gpurun_out/b/prof_c2/p_results.db > gpurun_out/b/c2_stats.csv 2> gpurun_out/b/c2_span.txt && rm -rf gpurun_out/b/prof_c2
timeout -k 10 300 python bench.py --no-modes --no-cpu-baseline > gpurun_out/b/bench_f16.json 2> gpurun_out/b/bench_f16.err || exit 1
timeout -k 10 300 python bench.py --no-modes --no-cpu-baseline --math bf16x3 > gpurun_out/b/bench_bf16.json 2> gpurun_out/b/bench_bf16.err || exit 1
tail -3 gpurun_out/b/tests.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/b/bench_*.json')):
    j=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f, j['value'], j['ms_per_step'])
PY
