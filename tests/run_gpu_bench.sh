# GPU box: parity tests, bench, rocprofv3 kernel trace of the bench.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/tests.log 2>&1 || (tail -60 gpurun_out/tests.log; exit 1)
tail -3 gpurun_out/tests.log
timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1 || (tail -40 gpurun_out/bench.log; exit 1)
tail -2 gpurun_out/bench.log
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1 || (tail -40 $GRAFT_REPO_ROOT/gpurun_out/prof.log; exit 1)
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof.log
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof | head -30
