set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || (tail -60 gpurun_out/gpu_tests.log; exit 1)
tail -3 gpurun_out/gpu_tests.log
timeout -k 10 120 python tools/partition_probe.py 1 2 4 8
RTX_SHARE_WORK=0 timeout -k 10 120 python tools/partition_probe.py 1 8
