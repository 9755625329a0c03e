# GPU box: the round-2 measurements that profiles/r02_* are copied from.  Stage $1:
#   a  default bench line, rocprofv3 --kernel-trace --stats of the same command, PMC passes of the headline frame
#   b  PMC passes of BASELINE config 5 (20M-triangle soup, 2048^2, 16 spp) and of the 4M soup
#   c  frame timelines (whole frame, 1/8 share), partition probe, the other configs, the load microbenchmarks
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
case "$1" in
a)
  bash tests/run_gpu_bench.sh || exit 1
  cp gpurun_out/bench.json gpurun_out/r02/bench.json
  cp gpurun_out/prof/bench_kernel_stats.csv gpurun_out/r02/kernel_stats.csv
  TAG=headline WKEY="scan:0:1024:16:4:1" BENCH_ARGS="" bash tests/run_gpu_pmc.sh > gpurun_out/r02/pmc_headline.txt 2>&1 || { tail -5 gpurun_out/r02/pmc_headline.txt; exit 1; }
  tail -3 gpurun_out/r02/pmc_headline.txt
  ;;
b)
  TAG=soup4m WKEY="soup:4000000:1024:8:4:1" BENCH_ARGS="--scene soup --tris 4000000 --width 1024 --spp 8" bash tests/run_gpu_pmc.sh > gpurun_out/r02/pmc_soup4m.txt 2>&1 || { tail -5 gpurun_out/r02/pmc_soup4m.txt; exit 1; }
  tail -1 gpurun_out/r02/pmc_soup4m.txt
  cp gpurun_out/pmc_soup4m/pmc.json profiles/pmc.json
  TAG=config5 WKEY="soup:0:2048:16:4:1" BENCH_ARGS="--scene soup --width 2048 --spp 16" bash tests/run_gpu_pmc.sh > gpurun_out/r02/pmc_config5.txt 2>&1 || { tail -5 gpurun_out/r02/pmc_config5.txt; exit 1; }
  tail -1 gpurun_out/r02/pmc_config5.txt
  ;;
c)
  bash tests/run_gpu_timeline.sh 1 > gpurun_out/r02/timeline_full.txt 2>&1 || { tail -5 gpurun_out/r02/timeline_full.txt; exit 1; }
  bash tests/run_gpu_timeline.sh 8 > gpurun_out/r02/timeline_eighth.txt 2>&1 || { tail -5 gpurun_out/r02/timeline_eighth.txt; exit 1; }
  timeout -k 10 200 python tools/partition_probe.py 1 2 4 8 > gpurun_out/r02/partition_probe.txt 2>&1 || { tail -5 gpurun_out/r02/partition_probe.txt; exit 1; }
  cat gpurun_out/r02/partition_probe.txt
  bash tests/run_gpu_configs.sh > gpurun_out/r02/configs.txt 2>&1 || { tail -5 gpurun_out/r02/configs.txt; exit 1; }
  cp gpurun_out/configs.log gpurun_out/r02/other_configs.jsonl
  cat gpurun_out/r02/configs.txt
  (timeout -k 10 60 tools/coop_ubench 46 && timeout -k 10 60 tools/coop_ubench 400 && timeout -k 10 60 tools/gather_ubench) > gpurun_out/r02/load_ubench.txt 2>&1 || { tail -5 gpurun_out/r02/load_ubench.txt; exit 1; }
  cat gpurun_out/r02/load_ubench.txt
  ;;
esac
