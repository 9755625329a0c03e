# GPU box: HBM-side traffic of the 20 M-triangle soup (config 5): FETCH_SIZE pass + kernel trace.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_soup
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_soup -o tcc2 -- python3 $R/bench.py --scene soup --width 2048 --spp 16 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_soup/tcc2.log 2>&1 || { tail -20 $R/gpurun_out/pmc_soup/tcc2.log; exit 1; }
tail -1 $R/gpurun_out/pmc_soup/tcc2.log | cut -c1-300
