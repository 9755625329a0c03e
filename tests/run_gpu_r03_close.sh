#!/bin/bash
# Round-3 closing sequence, one stage per GPU call (profiles/pmc.json travels back and forth between them):
#   tests  the whole GPU suite          a / b / c  tests/run_gpu_r03_profiles.sh stages
#   rest   PMC of configs 1, 2, 4, 4M soup          bench  smoke() + the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
case "$1" in
tests)
  timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03/gpu_tests.log 2>&1 || { tail -40 gpurun_out/r03/gpu_tests.log; exit 1; }
  tail -3 gpurun_out/r03/gpu_tests.log ;;
a|b|c) bash tests/run_gpu_r03_profiles.sh $1 || exit 1
  [ -f gpurun_out/pmc_headline/pmc.json ] && [ "$1" = a ] && cp gpurun_out/pmc_headline/pmc.json gpurun_out/r03/pmc_after_a.json
  [ -f gpurun_out/pmc_config5/pmc.json ] && [ "$1" = b ] && cp gpurun_out/pmc_config5/pmc.json gpurun_out/r03/pmc_after_b.json
  true ;;
rest) bash tests/run_gpu_r03_pmc_rest.sh || exit 1 ;;
bench)
  timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03/smoke.log 2>&1 || { tail -20 gpurun_out/r03/smoke.log; exit 1; }
  tail -2 gpurun_out/r03/smoke.log
  timeout -k 10 500 python -u bench.py --also-two-slots > gpurun_out/r03/bench.log 2>&1 || { tail -30 gpurun_out/r03/bench.log; exit 1; }
  tail -1 gpurun_out/r03/bench.log > gpurun_out/r03/bench.json
  cut -c1-400 gpurun_out/r03/bench.json ;;
esac
