# GPU box: rocprofv3 PMC passes over a short bench run (each pass its own run).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
run_pass () {
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc -o $name -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc/$name.log 2>&1 || { tail -20 $R/gpurun_out/pmc/$name.log; exit 1; }
}
if [ "$1" = "rdreq" ]; then
run_pass rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
exit 0
fi
if [ "$1" = "mem" ]; then
run_pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_LATENCY_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY && \
run_pass tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM_RD
ls $R/gpurun_out/pmc
exit 0
fi
run_pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY && \
run_pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM && \
run_pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum && \
run_pass tcc2 FETCH_SIZE && \
run_pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE GRBM_COUNT
ls $R/gpurun_out/pmc
