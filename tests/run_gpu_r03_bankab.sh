#!/bin/bash
# The leaf-banking kernel variant forced on / off (RTX_LEAF_BANK) on the scenes whose default is the other one.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 0 1; do
  for set in "headline --steps 8 --warmup 2" "glass --scene scan_glass --width 2048 --spp 8 --steps 2 --warmup 1" "spheres --scene random_spheres --tris 1000 --width 1024 --spp 1 --steps 10 --warmup 2" "soup4m --scene soup --tris 4000000 --width 1024 --spp 8 --steps 3 --warmup 1"; do
    set -- $set; n=$1; shift
    RTX_LEAF_BANK=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-reference-visits "$@" > gpurun_out/bank_${v}_$n.log 2>&1 || { echo "$v $n FAILED"; tail -5 gpurun_out/bank_${v}_$n.log; exit 1; }
    python - "bank=$v $n" gpurun_out/bank_${v}_$n.log <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[2]).read().strip().splitlines() if l.startswith('{')][-1])
r=d["roofline"]
print(f"{sys.argv[1]:24s} ms/step {d['ms_per_step']:8.3f}  path {r['avg_launch_ms']:8.3f}  resolve {r['resolve_ms_per_step']:6.3f}  Mrays/s {d['value']:9.1f}")
PY
  done
done
