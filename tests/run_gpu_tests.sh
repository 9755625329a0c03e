set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/first_tests.log 2>&1 || (tail -60 gpurun_out/first_tests.log; exit 1)
tail -5 gpurun_out/first_tests.log
