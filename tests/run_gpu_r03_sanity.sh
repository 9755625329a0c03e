#!/bin/bash
# Sanity run on a fresh box after a rebuild: the whole GPU suite, smoke(), the default bench line -> gpurun_out/r03s/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03s
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03s/gpu_tests.log 2>&1 || { tail -40 gpurun_out/r03s/gpu_tests.log; exit 1; }
tail -3 gpurun_out/r03s/gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03s/smoke.log 2>&1 || { tail -20 gpurun_out/r03s/smoke.log; exit 1; }
tail -2 gpurun_out/r03s/smoke.log
timeout -k 10 500 python -u bench.py --also-two-slots > gpurun_out/r03s/bench.log 2>&1 || { tail -30 gpurun_out/r03s/bench.log; exit 1; }
tail -1 gpurun_out/r03s/bench.log > gpurun_out/r03s/bench.json
cut -c1-1500 gpurun_out/r03s/bench.json
