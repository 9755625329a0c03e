# PMC A/B (SQ instruction counters + TCC hit/miss) of the round-1 tree against the current one on $BENCH_ARGS.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for side in old new; do
  rm -rf $R/gpurun_out/pmc_$side; mkdir -p $R/gpurun_out/pmc_$side
  if [ $side = old ]; then B=$R/old_r01/bench.py; X=""; else B=$R/bench.py; X="--no-reference-visits"; fi
  cd /tmp
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$side -o sq1 -- python3 $B --steps 1 --warmup 0 --no-cpu-baseline $X $BENCH_ARGS > $R/gpurun_out/pmc_$side/sq1.log 2>&1 || { tail -5 $R/gpurun_out/pmc_$side/sq1.log; exit 1; }
  timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$side -o tcc1 -- python3 $B --steps 1 --warmup 0 --no-cpu-baseline $X $BENCH_ARGS > $R/gpurun_out/pmc_$side/tcc1.log 2>&1 || { tail -5 $R/gpurun_out/pmc_$side/tcc1.log; exit 1; }
  timeout -s KILL 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$side -o tcp1 -- python3 $B --steps 1 --warmup 0 --no-cpu-baseline $X $BENCH_ARGS > $R/gpurun_out/pmc_$side/tcp1.log 2>&1 || { tail -5 $R/gpurun_out/pmc_$side/tcp1.log; exit 1; }
  cd $R
  echo "######## $side"
  python tests/pmc_summary.py gpurun_out/pmc_$side | grep -v "COUNT\|<true" | grep -A18 "== path_kernel<false\|== path_ovh_kernel<false, true\|== trace_\|== shade_forward"
done
