#!/bin/bash
# PMC passes of the workloads that run_gpu_r03_profiles.sh a / b do not cover (configs 1, 2, 4 and the 4 M soup);
# every pass of a workload updates the copy of profiles/pmc.json that the next one starts from.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
run () { # tag, workload key, bench args
  TAG=$1 WKEY="$2" BENCH_ARGS="$3" bash tests/run_gpu_pmc.sh > gpurun_out/r03/pmc_$1.txt 2>&1 || { tail -5 gpurun_out/r03/pmc_$1.txt; exit 1; }
  tail -1 gpurun_out/r03/pmc_$1.txt | cut -c1-200
  cp gpurun_out/pmc_$1/pmc.json profiles/pmc.json
  cp gpurun_out/pmc_$1/summary.txt gpurun_out/r03/pmc_$1_summary.txt
}
run config1 "random_spheres:1000:1024:1:4:1" "--scene random_spheres --tris 1000 --width 1024 --spp 1" && \
run config2 "basic_suzanne:0:1024:4:4:1" "--scene basic_suzanne --width 1024 --spp 4" && \
run config4 "scan_glass:0:4096:64:4:1" "--scene scan_glass --width 4096 --spp 64" && \
run soup4m "soup:4000000:1024:8:4:1" "--scene soup --tris 4000000 --width 1024 --spp 8"
cp profiles/pmc.json gpurun_out/r03/pmc_rest.json
