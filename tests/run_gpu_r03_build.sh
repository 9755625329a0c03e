#!/bin/bash
# GPU-assisted binning: its tests, then build times with and without it (bench lines carry bvh.build_s)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_build.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/build_tests.log 2>&1 || { tail -40 gpurun_out/build_tests.log; exit 1; }
tail -2 gpurun_out/build_tests.log
for gb in 0 1; do
  for set in "scan 0 1024 16" "soup 4000000 1024 8" "soup 0 2048 16"; do
    set -- $set
    RTX_GPU_BINNING=$gb timeout -k 10 300 python bench.py --no-cpu-baseline --no-reference-visits --scene $1 --tris $2 --width $3 --spp $4 --steps 2 --warmup 1 > gpurun_out/build_${gb}.log 2>&1 || { tail -5 gpurun_out/build_${gb}.log; exit 1; }
    python - "binning=$gb $1 $2" gpurun_out/build_${gb}.log <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[2]).read().strip().splitlines() if l.startswith('{')][-1])
b=d["config"]["bvh"]
print(f"{sys.argv[1]:28s} build {b['build_s']:.3f} s  flatten {b['flatten_s']:.3f} s  upload {b['upload_s']:.3f} s  ms/step {d['ms_per_step']:.3f}")
PY
  done
done
