#!/bin/bash
# Round-3 closing run on one GPU box: PMC passes of the 4 M-triangle soup (the last stale entry of profiles/pmc.json),
# the whole GPU suite, smoke(), and the default bench line -> gpurun_out/r03/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
TAG=soup4m WKEY="soup:4000000:1024:8:4:1" BENCH_ARGS="--scene soup --tris 4000000 --width 1024 --spp 8" bash tests/run_gpu_pmc.sh > gpurun_out/r03/pmc_soup4m.txt 2>&1 || { tail -5 gpurun_out/r03/pmc_soup4m.txt; exit 1; }
tail -1 gpurun_out/r03/pmc_soup4m.txt | cut -c1-300
cp gpurun_out/pmc_soup4m/pmc.json profiles/pmc.json
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03/gpu_tests.log 2>&1 || { tail -40 gpurun_out/r03/gpu_tests.log; exit 1; }
tail -3 gpurun_out/r03/gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03/smoke.log 2>&1 || { tail -20 gpurun_out/r03/smoke.log; exit 1; }
tail -2 gpurun_out/r03/smoke.log
timeout -k 10 500 python -u bench.py --also-two-slots > gpurun_out/r03/bench.log 2>&1 || { tail -30 gpurun_out/r03/bench.log; exit 1; }
tail -1 gpurun_out/r03/bench.log > gpurun_out/r03/bench.json
cp profiles/pmc.json gpurun_out/r03/pmc.json
cut -c1-1500 gpurun_out/r03/bench.json
