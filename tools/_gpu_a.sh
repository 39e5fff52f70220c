cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_planes.py tests/test_hip_ops.py -m gpu -x -q > gpurun_out/a_tests.log 2>&1 && \
timeout -k 10 400 python bench.py --no-modes --no-cpu-baseline > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err && \
timeout -k 10 400 python bench.py --no-modes --no-cpu-baseline --math bf16x3 > gpurun_out/a_bench_bf16.json 2> gpurun_out/a_bench_bf16.err
tail -3 gpurun_out/a_tests.log; python - <<'PY'
import json
for f in ('gpurun_out/a_bench.json','gpurun_out/a_bench_bf16.json'):
    try:
        j=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f, j['value'], j['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
