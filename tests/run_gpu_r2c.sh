set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_parity.log 2>&1 || (tail -60 gpurun_out/gpu_parity.log; exit 1)
tail -3 gpurun_out/gpu_parity.log
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r2c.log 2>&1 || (tail -30 gpurun_out/bench_r2c.log; exit 1)
tail -1 gpurun_out/bench_r2c.log | cut -c1-300
