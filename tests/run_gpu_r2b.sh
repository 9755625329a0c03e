set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_zz_no_alloc.py -m gpu -x -v --timeout 120 --timeout-method thread > gpurun_out/noalloc.log 2>&1 || (tail -60 gpurun_out/noalloc.log; exit 1)
tail -4 gpurun_out/noalloc.log
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || (tail -60 gpurun_out/gpu_tests.log; exit 1)
tail -3 gpurun_out/gpu_tests.log
