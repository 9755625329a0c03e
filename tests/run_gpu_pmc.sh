# PMC passes (one rocprofv3 run each: SQ, TCC, TCP, TCP/TA/TD stalls) of one bench frame on $BENCH_ARGS -> gpurun_out/pmc_$TAG
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-cur}
D=$R/gpurun_out/pmc_$TAG
rm -rf $D; mkdir -p $D
cd /tmp
pass () { # name, counters...
  n=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $D -o $n -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-reference-visits --slots 1 $BENCH_ARGS > $D/$n.log 2>&1 || { echo "pass $n failed"; tail -3 $D/$n.log; return 1; }
}
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD || exit 1
pass sq2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS || exit 1
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum || exit 1
pass tcc2 TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum || exit 1
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum || exit 1
pass tcp2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT || exit 1
pass fetch FETCH_SIZE || exit 1
pass write WRITE_SIZE || exit 1
cd $R
# $WKEY (bench.py's workload key, "scene:tris:width:spp:bounces:world") -> entry of that workload in gpurun_out/pmc_$TAG/pmc.json
if [ -n "$WKEY" ]; then cp profiles/pmc.json gpurun_out/pmc_$TAG/pmc.json 2>/dev/null; python tests/pmc_summary.py gpurun_out/pmc_$TAG gpurun_out/pmc_$TAG/pmc.json "$WKEY" "tests/run_gpu_pmc.sh, BENCH_ARGS='$BENCH_ARGS'" > gpurun_out/pmc_$TAG/summary.txt
else python tests/pmc_summary.py gpurun_out/pmc_$TAG > gpurun_out/pmc_$TAG/summary.txt; fi
grep -A48 "== path_kernel<false, false, true\|== path_ovh_kernel<false, true" gpurun_out/pmc_$TAG/summary.txt; tail -1 gpurun_out/pmc_$TAG/summary.txt
