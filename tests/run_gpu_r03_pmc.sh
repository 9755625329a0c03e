#!/bin/bash
# Round-3 PMC A/B: one SQ pass (+ GRBM) per library in gpurun_variants/ on $BENCH_ARGS (default: the headline frame).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for so in gpurun_variants/librtx_*.so; do
  n=$(basename $so .so)
  rm -rf $R/gpurun_out/pmc_$n; mkdir -p $R/gpurun_out/pmc_$n
  export RTX_LIB=$R/$so
  cd /tmp
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$n -o sq1 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-reference-visits $BENCH_ARGS > $R/gpurun_out/pmc_$n/sq1.log 2>&1 || { tail -5 $R/gpurun_out/pmc_$n/sq1.log; exit 1; }
  if [ "${CACHE:-0}" = 1 ]; then
  timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$n -o tcc1 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-reference-visits $BENCH_ARGS > $R/gpurun_out/pmc_$n/tcc1.log 2>&1 || { tail -5 $R/gpurun_out/pmc_$n/tcc1.log; exit 1; }
  fi
  if [ "${MORE:-0}" = 1 ]; then
  timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$n -o sq2 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-reference-visits $BENCH_ARGS > $R/gpurun_out/pmc_$n/sq2.log 2>&1 || { tail -5 $R/gpurun_out/pmc_$n/sq2.log; exit 1; }
  fi
  cd $R
  echo "######## $n"
  python tests/pmc_summary.py gpurun_out/pmc_$n | grep -v "COUNT\|<true" | grep -A${LINES:-22} "== path_kernel<false, false, true\|== path_ovh_kernel<false, true"
  python - $R/gpurun_out/pmc_$n <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+"/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "path_" in r["Kernel_Name"] and "Lb1ELb0" not in r["Kernel_Name"]:
            print("   launch", r["Kernel_Name"][:60], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, "ms")
PY
done
