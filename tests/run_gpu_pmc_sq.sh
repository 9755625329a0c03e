# GPU box: the two SQ passes only (instruction counts, active lanes) over one bench step.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
rm -f gpurun_out/pmc/*
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
run_pass () {
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc -o $name -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline $BENCH_ARGS > $R/gpurun_out/pmc/$name.log 2>&1 || { tail -20 $R/gpurun_out/pmc/$name.log; exit 1; }
}
run_pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY && \
run_pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM && \
run_pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE GRBM_COUNT
