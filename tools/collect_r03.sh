#!/bin/bash
# Copies what the GPU calls of tests/run_gpu_r03_close.sh left under gpurun_out/ into profiles/ under round-3 names.
cd "$(dirname "$0")/.."
G=gpurun_out; P=profiles
cpif () { [ -f "$1" ] && cp "$1" "$2"; }
# (the rows of the trace path's kernels only: the same passes also see the scene upload's build kernels and copies)
for n in sq1 sq2 tcc1 tcc2 tcp1 tcp2 grbm fetch write; do
  [ -f $G/pmc_headline/${n}_counter_collection.csv ] && { head -1 $G/pmc_headline/${n}_counter_collection.csv; grep -E '"(void )?rtx::\(anonymous namespace\)::(path_kernel|path_ovh_kernel|resolve_kernel|raygen_kernel|accumulate_kernel)' $G/pmc_headline/${n}_counter_collection.csv; } > $P/r03_pmc_headline_$n.csv
done
cpif $G/pmc_headline/summary.txt $P/r03_pmc_summary.txt
for t in config1 config2 config4 config5 soup4m; do cpif $G/pmc_$t/summary.txt $P/r03_pmc_${t}_summary.txt; done
cpif $G/r03/kernel_stats.csv $P/r03_kernel_stats.csv
cpif $G/r03/kernel_stats_config4.csv $P/r03_kernel_stats_config4.csv
cpif $G/r03/kernel_stats_config5.csv $P/r03_kernel_stats_config5.csv
cpif $G/r03/other_configs.jsonl $P/r03_other_configs.jsonl
if [ -f $G/r03/wave_times.txt ]; then
  { echo "# tests/run_gpu_r03_wt.sh: per-wave times of the production path kernel (diagnostic build -DRTX_WAVE_TIMES: three timestamps per wave, no other change), headline scene, whole frame and rank 0's 1/8 share, max_ray_bounces 1 and 4; two renders each"; cat $G/r03/wave_times.txt; } > $P/r03_wave_times.txt
fi
if [ -f $G/r03/partition_probe.txt ]; then
  { echo "# tools/partition_probe.py 1 2 4 8 16 -- rank 0 share of the headline frame on one MI355X, round-3 kernels; alternating slots = two render slots, then three and four in rotation"; cat $G/r03/partition_probe.txt; } > $P/r03_partition_probe.txt
fi
cpif $G/r03/bench.json $P/r03_bench.json
if [ -f $G/r03/valu_ubench.txt ]; then
  { echo "# tools/valu_ubench.hip on one MI355X: time per wave64 instruction per SIMD with 1 / 5 / 8 waves per SIMD (a block of 256 = one wave per SIMD; independent instructions, finite data); 'cycles' at an assumed 2.4 / 2.1 GHz -- the ns are the measurement.  The cswap rows execute 10 instructions where the others execute 16: their per-instruction figures are x 1.6"; cat $G/r03/valu_ubench.txt; } > $P/r03_valu_ubench.txt
fi
cpif $G/r03/gpu_tests.log $P/r03_gpu_tests.txt
if [ -f $G/r03/timeline_full.txt ] && [ -f $G/r03/timeline_eighth.txt ]; then
  { echo "# tests/run_gpu_timeline.sh 1 / 8: kernels of one warm frame (ms: start, end, duration), whole frame then rank 0's 1/8 share"; cat $G/r03/timeline_full.txt; echo; cat $G/r03/timeline_eighth.txt; } > $P/r03_frame_timeline.txt
fi
ls -la $P | grep r03 | wc -l
