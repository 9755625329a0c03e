# GPU box: the round-3 measurements that profiles/r03_* are copied from.  Stage $1:
#   a  default bench line, rocprofv3 --kernel-trace --stats of the same command, PMC passes of the headline frame
#   b  PMC passes of BASELINE config 5 (20M-triangle soup, 2048^2, 16 spp) and kernel stats of configs 4 and 5
#   c  partition probe, the other configs, per-wave times (diagnostic build), frame timelines
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
case "$1" in
a)
  bash tests/run_gpu_bench.sh || exit 1
  cp gpurun_out/bench.json gpurun_out/r03/bench.json
  cp gpurun_out/prof/bench_kernel_stats.csv gpurun_out/r03/kernel_stats.csv
  TAG=headline WKEY="scan:0:1024:16:4:1" BENCH_ARGS="" bash tests/run_gpu_pmc.sh > gpurun_out/r03/pmc_headline.txt 2>&1 || { tail -5 gpurun_out/r03/pmc_headline.txt; exit 1; }
  tail -3 gpurun_out/r03/pmc_headline.txt
  ;;
b)
  TAG=config5 WKEY="soup:0:2048:16:4:1" BENCH_ARGS="--scene soup --width 2048 --spp 16" bash tests/run_gpu_pmc.sh > gpurun_out/r03/pmc_config5.txt 2>&1 || { tail -5 gpurun_out/r03/pmc_config5.txt; exit 1; }
  tail -1 gpurun_out/r03/pmc_config5.txt
  export TMPDIR=/tmp
  for cfg in "config4 --scene scan_glass --width 4096 --spp 64 --steps 2 --warmup 1" "config5 --scene soup --width 2048 --spp 16 --steps 3 --warmup 1"; do
    set -- $cfg; tag=$1; shift
    rm -rf gpurun_out/prof_$tag
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o bench -- python3 $R/bench.py --no-cpu-baseline --no-reference-visits "$@" > $R/gpurun_out/prof_$tag.log 2>&1) || { tail -20 gpurun_out/prof_$tag.log; exit 1; }
    cp gpurun_out/prof_$tag/bench_kernel_stats.csv gpurun_out/r03/kernel_stats_$tag.csv
    head -4 gpurun_out/r03/kernel_stats_$tag.csv | cut -c1-160
  done
  ;;
c)
  timeout -k 10 250 python tools/partition_probe.py 1 2 4 8 16 > gpurun_out/r03/partition_probe.txt 2>&1 || { tail -5 gpurun_out/r03/partition_probe.txt; exit 1; }
  cat gpurun_out/r03/partition_probe.txt | cut -c1-260
  bash tests/run_gpu_configs.sh > gpurun_out/r03/configs.txt 2>&1 || { tail -5 gpurun_out/r03/configs.txt; exit 1; }
  cp gpurun_out/configs.log gpurun_out/r03/other_configs.jsonl
  cat gpurun_out/r03/configs.txt
  BOUNCES="1 4" PARTS="1 8" bash tests/run_gpu_r03_wt.sh > gpurun_out/r03/wave_times.txt 2>&1 || { tail -5 gpurun_out/r03/wave_times.txt; exit 1; }
  cat gpurun_out/r03/wave_times.txt | cut -c1-330
  bash tests/run_gpu_timeline.sh 1 > gpurun_out/r03/timeline_full.txt 2>&1 || { tail -5 gpurun_out/r03/timeline_full.txt; exit 1; }
  bash tests/run_gpu_timeline.sh 8 > gpurun_out/r03/timeline_eighth.txt 2>&1 || { tail -5 gpurun_out/r03/timeline_eighth.txt; exit 1; }
  tail -12 gpurun_out/r03/timeline_full.txt
  ;;
esac
