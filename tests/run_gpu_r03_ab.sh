#!/bin/bash
# Round-3 A/B on one GPU box: (optionally) the parity tests on the in-tree library, then every library in
# gpurun_variants/ on the headline frame, config 5 and config 2 (bench lines -> gpurun_out/ab_*.log).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "${RUN_TESTS:-1}" = 1 ]; then
  timeout -k 10 ${TEST_TIMEOUT:-420} python -u -m pytest ${TESTS:-tests/test_gpu_parity.py} -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/ab_tests.log 2>&1 || { tail -40 gpurun_out/ab_tests.log; exit 1; }
  tail -3 gpurun_out/ab_tests.log
fi
show () {
python - "$1" "$2" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[2]).read().strip().splitlines() if l.startswith('{')][-1])
r=d["roofline"]
print(f"{sys.argv[1]:34s} ms/step {d['ms_per_step']:8.3f}  path {r['avg_launch_ms']:8.3f}  resolve {r['resolve_ms_per_step']:6.3f}  Mrays/s {d['value']:9.1f}  nodes/ray {r['per_class']['closest']['nodes_per_ray']:.3f}/{r['per_class']['shadow']['nodes_per_ray']:.3f} max {r.get('max_records_one_ray')}")
PY
}
for so in gpurun_variants/librtx_*.so; do
  n=$(basename $so .so)
  for set in ${SETS:-headline soup}; do
    case $set in
      headline) args="--steps 8 --warmup 2";;
      soup) args="--scene soup --width 2048 --spp 16 --steps 2 --warmup 1";;
      suzanne) args="--scene basic_suzanne --width 1024 --spp 4 --steps 10 --warmup 2";;
      spheres) args="--scene random_spheres --tris 1000 --width 1024 --spp 1 --steps 10 --warmup 2";;
      glass) args="--scene scan_glass --width 2048 --spp 8 --steps 2 --warmup 1";;
    esac
    RTX_LIB=$GRAFT_REPO_ROOT/$so timeout -k 10 200 python bench.py --no-cpu-baseline --no-reference-visits $args > gpurun_out/ab_${n}_$set.log 2>&1 || { echo "$n $set FAILED"; tail -5 gpurun_out/ab_${n}_$set.log; exit 1; }
    show "$n $set" gpurun_out/ab_${n}_$set.log
  done
done
