cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/b/tests.log 2>&1 || { tail -30 gpurun_out/b/tests.log; exit 1; }
tail -3 gpurun_out/b/tests.log
bash tools/evidence.sh
