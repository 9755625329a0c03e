#!/bin/bash
# build time of the 20 M soup and the headline scene against the piece size of the GPU top phase
cd $GRAFT_REPO_ROOT
for tp in 64 256 1024 4096; do
  for set in "soup 0 2048 16" "scan 0 1024 16"; do
    set -- $set
    RTX_GPU_BINNING=1 RTX_TOP_PIECES=$tp timeout -k 10 300 python bench.py --no-cpu-baseline --no-reference-visits --scene $1 --tris $2 --width $3 --spp $4 --steps 1 --warmup 0 > gpurun_out/build_tp.log 2>&1 || { tail -5 gpurun_out/build_tp.log; exit 1; }
    python - "pieces=$tp $1" gpurun_out/build_tp.log <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[2]).read().strip().splitlines() if l.startswith('{')][-1])
b=d["config"]["bvh"]
print(f"{sys.argv[1]:28s} build {b['build_s']:.3f} s  flatten {b['flatten_s']:.3f} s")
PY
  done
done
