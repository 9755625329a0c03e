# PMC passes of the other BASELINE configs (1, 2, 4) -> entries in gpurun_out/pmc_cfg/pmc.json (start: profiles/pmc.json)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_cfg
cp profiles/pmc.json gpurun_out/pmc_cfg/pmc.json
run () { # tag, wkey, bench args
  TAG=$1 WKEY="$2" BENCH_ARGS="$3" bash tests/run_gpu_pmc.sh > gpurun_out/pmc_cfg/$1.txt 2>&1 || { tail -5 gpurun_out/pmc_cfg/$1.txt; exit 1; }
  cp gpurun_out/pmc_$1/pmc.json profiles/pmc.json   # (the box's copy: the next run starts from it)
  cp gpurun_out/pmc_$1/pmc.json gpurun_out/pmc_cfg/pmc.json
  cp gpurun_out/pmc_$1/summary.txt gpurun_out/pmc_cfg/summary_$1.txt
  tail -1 gpurun_out/pmc_cfg/$1.txt | cut -c1-400
}
run config1 "random_spheres:1000:1024:1:4:1" "--scene random_spheres --tris 1000 --width 1024 --spp 1" && \
run config2 "basic_suzanne:0:1024:4:4:1" "--scene basic_suzanne --width 1024 --spp 4" && \
run config4 "scan_glass:0:4096:64:4:1" "--scene scan_glass --width 4096 --spp 64"
